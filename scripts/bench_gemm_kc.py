#!/usr/bin/env python3
"""Layout study, part 2: the library's own K-contiguous GEMM (csrc/gemm_kc.hip, `cot_study_gemm_kc`) beside the NCHW
1x1 kernel of the same layer and the vendor GEMM in the K-contiguous layout, B = 80, bf16, device time per call.
    python scripts/bench_gemm_kc.py [iters]
"""
import ctypes
import sys

import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from cotnet_amd import _lib  # noqa: E402

L = _lib.lib()
L.cot_study_gemm_kc.restype = ctypes.c_int
P = lambda t: ctypes.c_void_p(t.data_ptr())  # noqa: E731
ITERS = int(sys.argv[1]) if len(sys.argv) > 1 else 30
dev = torch.device("cuda:0")
stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def timed(fn):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(ITERS):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / ITERS


print(f"{'layer (N80)':28s} {'own K-contig 64/128 rows':>26s} {'vendor K-contig':>16s} {'own NCHW':>10s}   TFLOP/s of the best own K-contig, max|err|")
for N, HW, K, Nn in [(80, 196, 1024, 256), (80, 196, 256, 1024), (80, 49, 2048, 512), (80, 49, 512, 2048), (80, 196, 256, 256), (80, 49, 512, 512)]:
    M = N * HW
    x = torch.randn(M, K, device=dev).bfloat16()
    w = (torch.randn(Nn, K, device=dev) / K ** 0.5).bfloat16()
    y = torch.empty(M, Nn, device=dev, dtype=torch.bfloat16)
    ref = x @ w.t()
    ts = []
    for tm in (64, 128):
        rc = L.cot_study_gemm_kc(P(x), P(w), P(y), M, Nn, K, tm, stream)
        assert rc == 0, rc
        torch.cuda.synchronize()
        err = (y.float() - ref.float()).abs().max().item()
        ts.append(timed(lambda: L.cot_study_gemm_kc(P(x), P(w), P(y), M, Nn, K, tm, stream)))
    wt = w.t().contiguous()
    tv = timed(lambda: torch.matmul(x, wt, out=y))
    xn = torch.randn(N, K, HW, device=dev).bfloat16()
    yn = torch.empty(N, Nn, HW, device=dev, dtype=torch.bfloat16)
    tn = timed(lambda: L.cot_conv1x1_forward(P(xn), None, K, P(w), None, P(yn), N, K, Nn, HW, 2, stream))
    best = min(ts)
    print(f"{K:5d} -> {Nn:5d}  HW {HW:4d}       {ts[0]:10.1f} / {ts[1]:8.1f} us {tv:13.1f} us {tn:7.1f} us   {2.0 * M * K * Nn / best / 1e6:7.0f}   {err:.3f}")


# weight gradient, channels-last (gemm_kc_wgrad + reduce) beside the NCHW kernel's (cot_conv1x1_backward_weight)
L.cot_study_conv1x1_nhwc_wgrad_workspace.restype = ctypes.c_size_t
L.cot_conv1x1_workspace.restype = ctypes.c_size_t
print(f"\n{'weight gradient (N80)':28s} {'own K-contig':>14s} {'own NCHW':>10s}   max|err| / max|ref|")
for N, HW, K, Nn in [(80, 196, 1024, 256), (80, 196, 256, 1024), (80, 49, 2048, 512), (80, 49, 512, 2048), (80, 196, 256, 256), (80, 49, 512, 512)]:
    M = N * HW
    x = torch.randn(M, K, device=dev).bfloat16()
    dy = torch.randn(M, Nn, device=dev).bfloat16()
    dw = torch.empty(Nn, K, device=dev, dtype=torch.bfloat16)
    ws = torch.empty(L.cot_study_conv1x1_nhwc_wgrad_workspace(M, K, Nn, 0), device=dev, dtype=torch.uint8)
    assert L.cot_study_conv1x1_nhwc_wgrad(P(x), P(dy), P(dw), P(ws), M, K, Nn, 0, stream) == 0
    torch.cuda.synchronize()
    ref = dy.float().t() @ x.float()
    err = (dw.float() - ref).abs().max().item() / ref.abs().max().item()
    tk = timed(lambda: L.cot_study_conv1x1_nhwc_wgrad(P(x), P(dy), P(dw), P(ws), M, K, Nn, 0, stream))
    xn = torch.randn(N, K, HW, device=dev).bfloat16()
    gn = torch.randn(N, Nn, HW, device=dev).bfloat16()
    wsn = torch.empty(max(L.cot_conv1x1_workspace(N, K, Nn, HW, 0), 256), device=dev, dtype=torch.uint8)
    tn = timed(lambda: L.cot_conv1x1_backward_weight(P(gn), P(xn), None, K, P(dw), None, P(wsn), N, K, Nn, HW, 2, stream))
    print(f"{K:5d} -> {Nn:5d}  HW {HW:4d}       {tk:11.1f} us {tn:7.1f} us   {err:.4f}")


# data gradient straight from the untransposed weight (gemm_kc_nn) beside the forward form on a transposed copy
print(f"\n{'data gradient (N80)':28s} {'NN form':>10s} {'TN form on W^T':>16s}")
for N, HW, K, Nn in [(80, 196, 256, 1024), (80, 196, 1024, 256), (80, 49, 512, 2048), (80, 49, 2048, 512)]:   # K = Co, Nn = Ci
    M = N * HW
    dy = torch.randn(M, K, device=dev).bfloat16()
    w = (torch.randn(K, Nn, device=dev) / K ** 0.5).bfloat16()    # the convolution's weight [Co][Ci]
    wt = w.t().contiguous()
    dx = torch.empty(M, Nn, device=dev, dtype=torch.bfloat16)
    assert L.cot_study_conv1x1_nhwc_dgrad(P(dy), P(w), P(dx), 0, M, Nn, K, Nn, Nn, 0, stream) == 0
    torch.cuda.synchronize()
    err = (dx.float() - dy.float() @ w.float()).abs().max().item()
    t1 = timed(lambda: L.cot_study_conv1x1_nhwc_dgrad(P(dy), P(w), P(dx), 0, M, Nn, K, Nn, Nn, 0, stream))
    t2 = timed(lambda: L.cot_study_gemm_kc(P(dy), P(wt), P(dx), M, Nn, K, 0, stream)) if Nn % 128 == 0 else float("nan")
    print(f"{K:5d} -> {Nn:5d}  HW {HW:4d}       {t1:7.1f} us {t2:13.1f} us   max|err| {err:.3f}")

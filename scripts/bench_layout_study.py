#!/usr/bin/env python
"""Layout study for the 1x1 convolutions: is a K-contiguous ([N*HW, C], NHWC) GEMM worth a layout change?

Per layer shape of CoTNet-50 at B = 80, bf16, device time per call (HIP events, rotating buffer sets = cold, one set = warm) of

  ours      the library's NCHW kernels through the C ABI: forward / data gradient / weight gradient (+ split reduce)
  blas-nchw the SAME memory layout handed to the vendor GEMM (torch.matmul -> hipBLASLt / rocBLAS): per image
            Y[n] (Co x HW) = W (Co x Ci) . X[n] (Ci x HW) as a strided-batched GEMM; dX[n] = W^T . dY[n]; dW = sum_n dY[n] . X[n]^T
            (baddbmm-free: one batched GEMM writing [N, Co, Ci] partials + a sum over n: what a library call costs in NCHW)
  blas-nhwc the K-contiguous layout: X2 [N*HW, Ci], Y2 = X2 . W^T, dX2 = dY2 . W, dW = dY2^T . X2  (one plain GEMM each)

The vendor GEMMs are the yardstick for "what a tuned kernel reaches in that layout", not a product path.  Also reported: the
cost of moving one activation tensor between the layouts (NCHW <-> NHWC transposition, torch .contiguous()), which every
stage boundary -- or every non-GEMM kernel that stays NCHW -- would pay.
"""
import argparse
import ctypes
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from cotnet_amd import _lib  # noqa: E402

SHAPES = [("s1 conv1   256->64  @56", 256, 64, 56), ("s1 conv3    64->256 @56", 64, 256, 56),
          ("s2 conv1   512->128 @28", 512, 128, 28), ("s2 conv3   128->512 @28", 128, 512, 28),
          ("s3 conv1  1024->256 @14", 1024, 256, 14), ("s3 conv1x1 256->256 @14", 256, 256, 14), ("s3 conv3   256->1024@14", 256, 1024, 14),
          ("s4 conv1  2048->512 @7", 2048, 512, 7), ("s4 conv1x1 512->512 @7", 512, 512, 7), ("s4 conv3   512->2048@7", 512, 2048, 7)]


def P(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=80)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--json", default="")
    args = ap.parse_args()
    L = _lib.lib()
    dev = torch.device("cuda")
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    BF = _lib.COT_BF16
    N = args.batch
    rows = []

    def timeit(fn, nsets):
        for i in range(3):
            fn(i % nsets)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        out = []
        for sets in (nsets, 1):  # cold (rotating sets), then warm (one set)
            e0.record()
            for i in range(args.iters):
                fn(i % sets)
            e1.record()
            torch.cuda.synchronize()
            out.append(e0.elapsed_time(e1) / args.iters * 1e3)
        return out

    for name, Ci, Co, H in SHAPES:
        HW = H * H
        act_bytes = 2 * N * HW * (Ci + Co)
        nsets = max(2, min(8, int(600e6 // act_bytes) + 1))
        S = []
        for _ in range(nsets):
            x = torch.randn(N, Ci, H, H, device=dev).bfloat16()
            gy = torch.randn(N, Co, H, H, device=dev).bfloat16()
            S.append(dict(x=x, gy=gy, y=torch.empty_like(gy), gx=torch.empty_like(x),
                          x2=x.permute(0, 2, 3, 1).reshape(N * HW, Ci).contiguous(), gy2=gy.permute(0, 2, 3, 1).reshape(N * HW, Co).contiguous()))
        w = (torch.randn(Co, Ci, device=dev) * Ci ** -0.5).bfloat16()
        gw = torch.empty_like(w)
        ws = torch.empty(max(int(L.cot_conv1x1_workspace(N, Ci, Co, HW, 0)), 256), dtype=torch.uint8, device=dev)
        res = {}
        res["ours fwd"] = timeit(lambda i: L.cot_conv1x1_forward(P(S[i]["x"]), None, Ci, P(w), None, P(S[i]["y"]), N, Ci, Co, HW, BF, st), nsets)
        res["ours dgrad"] = timeit(lambda i: L.cot_conv1x1_backward_data(P(S[i]["gy"]), P(w), P(S[i]["gx"]), None, Ci, 0, P(ws), N, Ci, Co, HW, BF, st), nsets)
        res["ours wgrad"] = timeit(lambda i: L.cot_conv1x1_backward_weight(P(S[i]["gy"]), P(S[i]["x"]), None, Ci, P(gw), None, P(ws), N, Ci, Co, HW, BF, st), nsets)
        wt = w.t().contiguous()
        res["blas-nchw fwd"] = timeit(lambda i: torch.matmul(w, S[i]["x"].view(N, Ci, HW), out=S[i]["y"].view(N, Co, HW)), nsets)
        res["blas-nchw dgrad"] = timeit(lambda i: torch.matmul(wt, S[i]["gy"].view(N, Co, HW), out=S[i]["gx"].view(N, Ci, HW)), nsets)
        part = torch.empty(N, Co, Ci, device=dev, dtype=torch.bfloat16)
        res["blas-nchw wgrad"] = timeit(lambda i: (torch.bmm(S[i]["gy"].view(N, Co, HW), S[i]["x"].view(N, Ci, HW).transpose(1, 2), out=part), part.sum(0)), nsets)
        y2, gx2 = torch.empty(N * HW, Co, device=dev, dtype=torch.bfloat16), torch.empty(N * HW, Ci, device=dev, dtype=torch.bfloat16)
        res["blas-nhwc fwd"] = timeit(lambda i: torch.matmul(S[i]["x2"], wt, out=y2), nsets)
        res["blas-nhwc dgrad"] = timeit(lambda i: torch.matmul(S[i]["gy2"], w, out=gx2), nsets)
        res["blas-nhwc wgrad"] = timeit(lambda i: torch.matmul(S[i]["gy2"].t(), S[i]["x2"], out=gw), nsets)
        res["transpose x NCHW->NHWC"] = timeit(lambda i: S[i]["x"].permute(0, 2, 3, 1).contiguous(), nsets)
        row = dict(shape=name, act_MB=round(act_bytes / 1e6, 1), us={k: [round(v[0], 1), round(v[1], 1)] for k, v in res.items()})
        rows.append(row)
        hb = lambda us: act_bytes / (us * 1e-6) / 8e12  # noqa: E731
        print(f"{name:26s} {act_bytes / 1e6:6.1f} MB | " + " | ".join(
            f"{k.split()[0][:9]:9s} {k.split()[-1]:5s} {v[0]:6.1f}/{v[1]:6.1f}us {hb(v[1]):4.0%}" for k, v in res.items() if "transpose" not in k), flush=True)
        print(f"{'':26s}          transpose of x: {res['transpose x NCHW->NHWC'][0]:.1f}/{res['transpose x NCHW->NHWC'][1]:.1f} us", flush=True)
        del S
        torch.cuda.empty_cache()
    if args.json:
        json.dump(rows, open(args.json, "w"), indent=1)


if __name__ == "__main__":
    main()

#!/usr/bin/env python
"""3x3 / stride-2 poolings of a CoTNet-50 step through the C ABI (B = 80, bf16): plane-tile form (cot_set_tuning(27, 1), default)
against one lane per pixel (27, 0); time per launch and % of the 8 TB/s roofline for the bytes the op must move.

    python scripts/bench_pool.py"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from cotnet_amd import _lib  # noqa: E402


def main():
    L = _lib.lib()
    dev = torch.device("cuda:0")
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    P = lambda t: ctypes.c_void_p(t.data_ptr())
    N, BF = 80, _lib.COT_BF16

    def timed(fn, iters=20):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / iters * 1e3

    print(f"{'op':34s} {'per-pixel us':>13s} {'plane-tile us':>14s} {'%HBM':>6s}  identical")
    for name, C, H in (("maxpool 64ch 112->56", 64, 112), ("avgpool 128ch 56->28", 128, 56), ("avgpool 256ch 28->14", 256, 28),
                       ("avgpool 512ch 14->7", 512, 14)):
        Ho = (H - 1) // 2 + 1
        xs = [torch.relu(torch.randn(N, C, H, H, device=dev)).bfloat16() for _ in range(3)]
        gys = [torch.randn(N, C, Ho, Ho, device=dev).bfloat16() for _ in range(3)]
        y, gx = torch.empty_like(gys[0]), torch.empty_like(xs[0])
        taps = torch.empty(N, C, Ho, Ho, dtype=torch.uint8, device=dev)
        it = [0]
        if name.startswith("max"):
            def fwd():
                it[0] += 1
                assert L.cot_maxpool3x3s2_forward_taps(P(xs[it[0] % 3]), P(y), P(taps), N * C, H, H, BF, st) == 0

            def bwd():
                it[0] += 1
                assert L.cot_maxpool3x3s2_backward_taps(P(gys[it[0] % 3]), P(taps), P(gx), N * C, H, H, BF, st) == 0
            fb = xs[0].numel() * 2 + y.numel() * 3
            bb = y.numel() * 3 + gx.numel() * 2
        else:
            def fwd():
                it[0] += 1
                assert L.cot_avgpool3x3s2_forward(P(xs[it[0] % 3]), P(y), N * C, H, H, BF, st) == 0

            def bwd():
                it[0] += 1
                assert L.cot_avgpool3x3s2_backward(P(gys[it[0] % 3]), P(gx), N * C, H, H, BF, st) == 0
            fb = xs[0].numel() * 2 + y.numel() * 2
            bb = y.numel() * 2 + gx.numel() * 2
        for tag, fn, nb, outs in ((" fwd", fwd, fb, lambda: (y.clone(), taps.clone())), (" bwd", bwd, bb, lambda: (gx.clone(),))):
            res, t = [], []
            for tile in (0, 1):
                assert L.cot_set_tuning(27, tile) == 0
                it[0] = 0
                if tag == " bwd" and name.startswith("max"):
                    L.cot_maxpool3x3s2_forward_taps(P(xs[1]), P(y), P(taps), N * C, H, H, BF, st)
                fn()
                torch.cuda.synchronize()
                res.append(outs())
                t.append(timed(fn))
            same = all(torch.equal(a, b) for a, b in zip(*res))
            print(f"{name + tag:34s} {t[0]:13.1f} {t[1]:14.1f} {nb / (t[1] * 1e-6) / 8e12 * 100:6.1f}  {same}", flush=True)
    L.cot_set_tuning(27, 1)


if __name__ == "__main__":
    main()

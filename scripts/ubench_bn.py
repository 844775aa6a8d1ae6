#!/usr/bin/env python
"""BatchNorm+ReLU forward / backward through the C ABI on the CoTNet-50 tensor shapes (B = 80): streaming kernels
(cot_set_tuning(21, 0), finalize folded) vs channel-resident kernels (21 = 1).  Cold buffers (rotating sets > 256 MiB)."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cotnet_amd import _lib  # noqa: E402

SHAPES = [(64, 56), (256, 56), (128, 28), (512, 28), (64, 28), (256, 14), (128, 14), (1024, 14), (512, 7), (256, 7), (2048, 7)]


MODES = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "0,1").split(",")]


def P(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def main():
    L = _lib.lib()
    dev = torch.device("cuda:0")
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    N, BF = 80, _lib.COT_BF16
    L.cot_set_tuning(12, 1)
    print(f"{'C x HxH':14s} {'MB':>7s} | fwd us for cot_set_tuning(21, m), m = {MODES} (0 streaming, 1 channel-resident, >1 forced lanes) | bwd us")
    for C, H in SHAPES:
        HW = H * H
        nbytes = N * C * HW * 2
        nset = max(2, min(8, int(300e6 // (3 * nbytes)) + 1))
        sets = [(torch.randn(N, C, HW, device=dev).bfloat16(), torch.empty(N, C, HW, device=dev).bfloat16(),
                 torch.randn(N, C, HW, device=dev).bfloat16()) for _ in range(nset)]
        dx = torch.empty(N, C, HW, device=dev).bfloat16()
        gamma, beta = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev) * 0.1
        mean, rstd, dg, db = (torch.empty(C, device=dev) for _ in range(4))
        ws = torch.empty(int(L.cot_bn_act_workspace(N, C)), device=dev)
        res = {}
        for chan in MODES:
            L.cot_set_tuning(21, chan)

            def fwd(i):
                x, y, dy = sets[i % nset]
                rc = L.cot_bn_act_forward(P(x), None, P(y), P(gamma), P(beta), P(mean), P(rstd), None, None, None, P(ws), N, C, HW,
                                          1e-5, 0.1, 1, BF, st)
                assert rc == 0, L.cot_last_error()

            def bwd(i):
                x, y, dy = sets[i % nset]
                rc = L.cot_bn_act_backward(P(dy), P(x), None, P(dx), None, P(gamma), P(beta), P(mean), P(rstd), P(dg), P(db), P(ws),
                                           N, C, HW, 1, BF, st)
                assert rc == 0, L.cot_last_error()
            for fn in (fwd, bwd):
                for i in range(3):
                    fn(i)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for i in range(16):
                    fn(i)
                e1.record()
                torch.cuda.synchronize()
                res[(chan, fn.__name__)] = e0.elapsed_time(e1) / 16 * 1e3
        L.cot_set_tuning(21, 1)
        print(f"{C:5d} x {H:2d}x{H:<2d} {nbytes / 1e6:7.1f} | " + " ".join(f"{res[(m, 'fwd')]:9.1f}" for m in MODES) + " | " +
              " ".join(f"{res[(m, 'bwd')]:9.1f}" for m in MODES), flush=True)


if __name__ == "__main__":
    main()

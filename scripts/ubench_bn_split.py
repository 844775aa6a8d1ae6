#!/usr/bin/env python
"""Streaming BatchNorm+ReLU kernels through the C ABI (B = 80, bf16) as a function of the number of workgroups they aim for
(cot_set_tuning(28, target): channels x batch chunks), next to the channel-resident kernels where those exist.  Cold buffers.

    python scripts/ubench_bn_split.py"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cotnet_amd import _lib  # noqa: E402

SHAPES = [(64, 56), (32, 56), (256, 56), (128, 28), (64, 28), (512, 28), (64, 112)]
TARGETS = [512, 1024, 2048, 4096, 8192, 20480]


def P(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def main():
    L = _lib.lib()
    dev = torch.device("cuda:0")
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    N, BF = 80, _lib.COT_BF16
    print(f"{'C x HxH':12s} {'MB':>6s} | fwd us: chan-resident, then streaming at targets {TARGETS} | bwd us: same")
    for C, H in SHAPES:
        HW = H * H
        nbytes = N * C * HW * 2
        nset = max(2, min(8, int(300e6 // (3 * nbytes)) + 1))
        sets = [(torch.randn(N, C, HW, device=dev).bfloat16(), torch.empty(N, C, HW, device=dev).bfloat16(),
                 torch.randn(N, C, HW, device=dev).bfloat16()) for _ in range(nset)]
        dx = torch.empty(N, C, HW, device=dev).bfloat16()
        gamma, beta = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev) * 0.1
        mean, rstd, dg, db = (torch.empty(C, device=dev) for _ in range(4))
        res = {}
        for mode, target in [(1, 1024)] + [(0, t) for t in TARGETS]:
            assert L.cot_set_tuning(21, mode) == 0 and L.cot_set_tuning(28, target) == 0
            ws = torch.empty(int(L.cot_bn_act_workspace(N, C)), device=dev)

            def fwd(i):
                x, y, dy = sets[i % nset]
                assert L.cot_bn_act_forward(P(x), None, P(y), P(gamma), P(beta), P(mean), P(rstd), None, None, None, P(ws), N, C, HW,
                                            1e-5, 0.1, 1, BF, st) == 0, L.cot_last_error()

            def bwd(i):
                x, y, dy = sets[i % nset]
                assert L.cot_bn_act_backward(P(dy), P(x), None, P(dx), None, P(gamma), P(beta), P(mean), P(rstd), P(dg), P(db), P(ws),
                                             N, C, HW, 1, BF, st) == 0, L.cot_last_error()
            for fn in (fwd, bwd):
                for i in range(3):
                    fn(i)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for i in range(12):
                    fn(i)
                e1.record()
                torch.cuda.synchronize()
                res[(mode, target, fn.__name__)] = e0.elapsed_time(e1) / 12 * 1e3
        L.cot_set_tuning(21, 1)
        L.cot_set_tuning(28, 1024)
        keys = [(1, 1024)] + [(0, t) for t in TARGETS]
        print(f"{C:4d} x {H:3d}^2 {nbytes / 1e6:6.1f} | " + " ".join(f"{res[k + ('fwd',)]:7.1f}" for k in keys) + " | " +
              " ".join(f"{res[k + ('bwd',)]:7.1f}" for k in keys), flush=True)


if __name__ == "__main__":
    main()

#!/usr/bin/env python
"""GroupNorm-9 forward / backward through the C ABI at the CoTNet-50 stage shapes (B = 80, bf16), rotating buffers;
    python scripts/bench_gn9.py [iters] [49=0 49=1 ...]   (cot_set_tuning settings, one run each)"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from cotnet_amd import _lib  # noqa: E402

L = _lib.lib()
ITERS = int(sys.argv[1]) if len(sys.argv) > 1 else 30
TUNES = [a for a in sys.argv[2:] if "=" in a] or [None]
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
BF, B = _lib.COT_BF16, 80


def P(t):
    return ctypes.c_void_p(t.data_ptr())


def timed(fn):
    for i in range(3):
        fn(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(ITERS):
        fn(i)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / ITERS * 1e3


print(f"{'shape':22s} {'tuning':>8s} {'fwd us':>8s} {'bwd us':>8s}")
for G, HW in ((8, 3136), (16, 784), (32, 196), (64, 49)):
    C = 9 * G
    sets = [(torch.randn(B, C, HW, device="cuda").bfloat16(), torch.randn(B, C, HW, device="cuda").bfloat16(),
             torch.empty(B, C, HW, device="cuda").bfloat16(), torch.empty(B, C, HW, device="cuda").bfloat16()) for _ in range(4)]
    gamma, beta = torch.ones(C, device="cuda").bfloat16(), torch.zeros(C, device="cuda").bfloat16()
    mean, rstd = torch.empty(B * G, device="cuda"), torch.empty(B * G, device="cuda")
    dg, db = torch.empty(C, device="cuda").bfloat16(), torch.empty(C, device="cuda").bfloat16()
    ws = torch.empty(2 * B * C, device="cuda")
    for tune in TUNES:
        if tune:
            assert L.cot_set_tuning(int(tune.split("=")[0]), int(tune.split("=")[1])) == 0

        def f(i):
            x, dy, y, dx = sets[i % 4]
            assert L.cot_group_norm9_forward(P(x), P(gamma), P(beta), P(y), P(mean), P(rstd), B, C, HW, ctypes.c_float(1e-5), BF, st) == 0

        def bw(i):
            x, dy, y, dx = sets[i % 4]
            assert L.cot_group_norm9_backward(P(dy), P(x), P(mean), P(rstd), P(gamma), P(dx), P(dg), P(db), P(ws), B, C, HW, BF, st) == 0

        print(f"N80 G{G:<3d} HW{HW:<5d}       {tune or '-':>8s} {timed(f):8.1f} {timed(bw):8.1f}", flush=True)

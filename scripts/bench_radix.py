#!/usr/bin/env python
"""Radix-2 tail kernels (channel-major descriptors) through the C ABI at the CoTNet-50 stage shapes (B = 80, bf16), rotating buffers;
    python scripts/bench_radix.py [iters] [50=0 50=1 ...]   (cot_set_tuning settings, one run each)"""
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


print(f"{'shape':22s} {'tuning':>8s} {'gap_t':>7s} {'mix':>7s} {'bwd_red':>8s} {'bwd_app':>8s}   us")
for C, HW in ((64, 3136), (128, 784), (256, 196), (512, 49)):
    r = lambda: torch.randn(B, C, HW, device="cuda").bfloat16()  # noqa: E731
    sets = [(r(), r(), r(), torch.empty(B, C, HW, device="cuda").bfloat16(), torch.empty(B, C, HW, device="cuda").bfloat16()) for _ in range(4)]
    gapT = torch.empty(C, B, device="cuda").bfloat16()
    logT, glogT = torch.randn(2 * C, B, device="cuda").bfloat16(), torch.empty(2 * C, B, device="cuda").bfloat16()
    attn, ggapT = torch.empty(B * C * 2, device="cuda").bfloat16(), torch.randn(C, B, device="cuda").bfloat16()
    for tune in TUNES:
        if tune:
            assert L.cot_set_tuning(int(tune.split("=")[0]), int(tune.split("=")[1])) == 0

        def f1(i):
            y, k, g, o1, o2 = sets[i % 4]
            assert L.cot_radix_gap_t(P(y), P(k), P(gapT), B, C, HW, BF, st) == 0

        def f2(i):
            y, k, g, o1, o2 = sets[i % 4]
            assert L.cot_radix_mix_logits(P(y), P(k), P(logT), P(o1), P(attn), B, C, HW, BF, st) == 0

        def f3(i):
            y, k, g, o1, o2 = sets[i % 4]
            assert L.cot_radix_mix_backward_reduce(P(g), P(y), P(k), P(attn), P(glogT), B, C, HW, BF, st) == 0

        def f4(i):
            y, k, g, o1, o2 = sets[i % 4]
            assert L.cot_radix_mix_backward_apply(P(g), P(attn), P(ggapT), P(o1), P(o2), B, C, HW, BF, st) == 0

        print(f"N80 C{C:<4d} HW{HW:<5d}      {tune or '-':>8s} {timed(f1):7.1f} {timed(f2):7.1f} {timed(f3):8.1f} {timed(f4):8.1f}", flush=True)

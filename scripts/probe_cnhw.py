#!/usr/bin/env python
"""Layout probe, round 5: the 14x14 / 7x7 stages stored [C][N][HW] ("channel-major batch-inner": a channel's N planes are ONE
contiguous row of N*HW elements) need no new GEMM / BatchNorm kernels -- the existing NCHW entry points called with N = 1 and
HW' = N*HW compute exactly that layout.  This script times both forms of every 1x1 convolution and BatchNorm of stages 3 / 4 at
B = 80 (bf16) through the C ABI on rotating buffer sets: what would the layout buy before any plane kernel (3x3, aggregation,
GroupNorm, radix tail) learns the two strides?
    python scripts/probe_cnhw.py [iters]
"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from cotnet_amd import _lib  # noqa: E402

L = _lib.lib()
ITERS = int(sys.argv[1]) if len(sys.argv) > 1 else 30
TUNES = [a for a in sys.argv[2:] if "=" in a]  # e.g. 46=0 46=256 46=512: the channel-major form once per setting (cot_set_tuning)
dev = torch.device("cuda:0")
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
BF = _lib.COT_BF16
B = 80


def P(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


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


CONVS = [("s3 conv1  1024->256 @14", 1024, 256, 196, 0, False), ("s3 embed0  512->128 @14", 512, 128, 196, 256, False),
         ("s3 embed3  128->288 @14", 128, 288, 196, 0, True), ("s3 conv1x1 256->256 @14", 256, 256, 196, 0, False),
         ("s3 conv3   256->1024@14", 256, 1024, 196, 0, False), ("s4 conv1  2048->512 @7", 2048, 512, 49, 0, False),
         ("s4 embed0 1024->256 @7", 1024, 256, 49, 512, False), ("s4 embed3  256->576 @7", 256, 576, 49, 0, True),
         ("s4 conv1x1 512->512 @7", 512, 512, 49, 0, False), ("s4 conv3   512->2048@7", 512, 2048, 49, 0, False),
         ("s2 conv1   512->128 @28", 512, 128, 784, 0, False), ("s2 conv3   128->512 @28", 128, 512, 784, 0, False)]
print(f"{'1x1 convolution':26s} {'form':>14s} {'fwd us':>8s} {'dgrad us':>9s} {'wgrad us':>9s}")
tot = {"nchw": [0, 0, 0], "cnhw": [0, 0, 0]}
for name, Ci, Co, HW, split, bias in CONVS:
    forms = [("nchw", (B, HW), None)] + ([("cnhw " + t, (1, B * HW), t) for t in TUNES] or [("cnhw", (1, B * HW), None)])
    for form, (N, hw), tune in forms:
        for kv in (tune.split("+") if tune else []):  # ("48=2+49=4": several keys per variant)
            assert L.cot_set_tuning(int(kv.split("=")[0]), int(kv.split("=")[1])) == 0
        nset = max(2, min(6, int(300e6 // ((Ci + Co) * N * hw * 2)) + 1))
        sets = []
        for _ in range(nset):
            x1 = torch.randn(N, split or Ci, hw, device=dev).bfloat16()
            x2 = torch.randn(N, Ci - split, hw, device=dev).bfloat16() if split else None
            sets.append((x1, x2, torch.randn(N, Co, hw, device=dev).bfloat16(), torch.empty(N, Co, hw, device=dev).bfloat16(),
                         torch.empty_like(x1), torch.empty_like(x2) if split else None))
        w = (torch.randn(Co, Ci, device=dev) / Ci ** 0.5).bfloat16()
        b = torch.randn(Co, device=dev).bfloat16() if bias else None
        gw, gb = torch.empty_like(w), (torch.empty_like(b) if bias else None)
        ws = torch.empty(int(L.cot_conv1x1_workspace(N, Ci, Co, hw, 1 if bias else 0)), dtype=torch.uint8, device=dev)
        c1 = split or Ci

        def fwd(i):
            x1, x2, gy, y, g1, g2 = sets[i % nset]
            assert L.cot_conv1x1_forward(P(x1), P(x2), c1, P(w), P(b), P(y), N, Ci, Co, hw, BF, st) == 0, L.cot_last_error()

        def dgrad(i):
            x1, x2, gy, y, g1, g2 = sets[i % nset]
            assert L.cot_conv1x1_backward_data(P(gy), P(w), P(g1), P(g2), c1, 0, P(ws), N, Ci, Co, hw, BF, st) == 0, L.cot_last_error()

        def wgrad(i):
            x1, x2, gy, y, g1, g2 = sets[i % nset]
            assert L.cot_conv1x1_backward_weight(P(gy), P(x1), P(x2), c1, P(gw), P(gb), P(ws), N, Ci, Co, hw, BF, st) == 0, L.cot_last_error()

        try:
            t = [timed(fwd), timed(dgrad), timed(wgrad)]
        except AssertionError as e:
            print(f"{name:26s} {form:>14s}  unsupported: {e}")
            continue
        for k in range(3):
            tot.setdefault(form, [0, 0, 0])[k] += t[k]
        print(f"{name:26s} {form:>14s} {t[0]:8.1f} {t[1]:9.1f} {t[2]:9.1f}", flush=True)
print("sum", {k: [round(x, 1) for x in v] for k, v in tot.items()})

L.cot_bn_act_workspace.restype = ctypes.c_int
print(f"\n{'BatchNorm (+ReLU)':26s} {'form':>14s} {'fwd us':>8s} {'bwd us':>8s}")
for C, HW in ((256, 196), (128, 196), (1024, 196), (512, 49), (256, 49), (2048, 49), (128, 784), (512, 784)):
    for form, (N, hw) in (("nchw", (B, HW)), ("cnhw", (1, B * HW))):
        nset = max(2, min(6, int(300e6 // (3 * C * N * hw * 2)) + 1))
        sets = [(torch.randn(N, C, hw, device=dev).bfloat16(), torch.empty(N, C, hw, device=dev).bfloat16(),
                 torch.randn(N, C, hw, device=dev).bfloat16(), torch.empty(N, C, hw, device=dev).bfloat16()) for _ in range(nset)]
        g, bt = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev)
        mean, rstd, dg, db = (torch.empty(C, device=dev) for _ in range(4))
        ws = torch.empty(max(1, int(L.cot_bn_act_workspace(N, C))), device=dev)

        def f(i):
            x, y, dy, dx = sets[i % nset]
            assert L.cot_bn_act_forward(P(x), None, P(y), P(g), P(bt), P(mean), P(rstd), None, None, None, P(ws), N, C, hw,
                                        ctypes.c_float(1e-5), ctypes.c_float(0.1), 1, BF, st) == 0, L.cot_last_error()

        def bw(i):
            x, y, dy, dx = sets[i % nset]
            assert L.cot_bn_act_backward(P(dy), P(x), None, P(dx), None, P(g), P(bt), P(mean), P(rstd), P(dg), P(db), P(ws), N, C, hw, 1,
                                         BF, st) == 0, L.cot_last_error()

        try:
            print(f"N80 C{C:<5d} HW{HW:<14d} {form:>14s} {timed(f):8.1f} {timed(bw):8.1f}", flush=True)
        except AssertionError as e:
            print(f"N80 C{C:<5d} HW{HW:<14d} {form:>14s}  unsupported: {e}")

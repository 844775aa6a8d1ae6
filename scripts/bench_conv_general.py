#!/usr/bin/env python
"""Device time of the general convolution kernels (csrc/conv_gen.hip) next to MIOpen's for the same nn.Conv2d: forward and
forward+backward per layer shape, fp32 (the reference's precision) and the grouped / off-grid bf16 shapes of CoXtLayer.
    python scripts/bench_conv_general.py [--batch 80] [--iters 10] [--json out.json]
"""
import argparse
import json
import os
import sys

import torch
from torch import nn

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cotnet_amd import conv1x1 as c1, conv3x3g as c3  # noqa: E402

# (name, Ci, Co, groups, ksize, H, dtype)
SHAPES = [
    ("s1 conv1 256->64 fp32", 256, 64, 1, 1, 56, torch.float32),
    ("s1 key3x3 64 g4 fp32", 64, 64, 4, 3, 56, torch.float32),
    ("s1 conv3 64->256 fp32", 64, 256, 1, 1, 56, torch.float32),
    ("s3 conv1 1024->256 fp32", 1024, 256, 1, 1, 14, torch.float32),
    ("s3 key3x3 256 g4 fp32", 256, 256, 4, 3, 14, torch.float32),
    ("s4 conv3 512->2048 fp32", 512, 2048, 1, 1, 7, torch.float32),
    ("x1 embed0 192->48 g2 bf16", 192, 48, 2, 1, 56, torch.bfloat16),
    ("x1 embed3 48->108 g2 bf16", 48, 108, 2, 1, 56, torch.bfloat16),
    ("x1 conv1x1 96->96 g2 bf16", 96, 96, 2, 1, 56, torch.bfloat16),
    ("x1 key3x3 96 g8 bf16", 96, 96, 8, 3, 56, torch.bfloat16),
    ("x3 conv1x1 384 g2 bf16", 384, 384, 2, 1, 14, torch.bfloat16),
    ("x3 key3x3 384 g8 bf16", 384, 384, 8, 3, 14, torch.bfloat16),
]


def timed(fn, iters):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3  # us


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=80)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--json", default="")
    a = ap.parse_args()
    dev = "cuda"
    rows = []
    print(f"{'shape':30s} {'GFLOP':>7s} | {'hip fwd us':>10s} {'TF/s':>6s} {'hip f+b us':>10s} {'TF/s':>6s} | {'miopen fwd':>10s} {'miopen f+b':>10s}")
    for name, Ci, Co, G, k, H, dtype in SHAPES:
        conv = nn.Conv2d(Ci, Co, k, padding=k // 2, groups=G, bias=False).to(dev).to(dtype)
        x = torch.randn(a.batch, Ci, H, H, device=dev).to(dtype).requires_grad_(True)
        gy = torch.randn(a.batch, Co, H, H, device=dev).to(dtype)
        op = c1.conv1x1 if k == 1 else c3.conv3x3
        mod = c1 if k == 1 else c3
        gflop = 2.0 * a.batch * H * H * Co * (Ci // G) * k * k / 1e9
        res = {}
        for mode in ("hip", ""):
            mod.MODE = mode

            def fwd():
                with torch.no_grad():
                    return op(conv, x)

            def fb():
                conv.zero_grad(set_to_none=True)
                x.grad = None
                op(conv, x).backward(gy)
            res[mode or "miopen"] = (timed(fwd, a.iters), timed(fb, a.iters))
        mod.MODE = ""
        h, m = res["hip"], res["miopen"]
        tf = lambda us, mult=1.0: mult * gflop / us * 1e3  # GFLOP per us = 1000 TFLOP/s
        print(f"{name:30s} {gflop:7.2f} | {h[0]:10.1f} {tf(h[0]):6.1f} {h[1]:10.1f} {tf(h[1], 3):6.1f} | {m[0]:10.1f} {m[1]:10.1f}")
        rows.append({"shape": name, "gflop_fwd": gflop, "hip_fwd_us": h[0], "hip_fwd_bwd_us": h[1], "miopen_fwd_us": m[0],
                     "miopen_fwd_bwd_us": m[1], "hip_fwd_tflops": tf(h[0]), "hip_fwd_bwd_tflops": tf(h[1], 3)})
    if a.json:
        json.dump(rows, open(a.json, "w"), indent=1)


if __name__ == "__main__":
    main()

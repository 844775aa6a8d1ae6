#!/usr/bin/env python
"""A/B of the three 1x1-convolution implementations at the CoTNet-50 layer shapes (B=80, bf16):
   module = nn.Conv2d (MIOpen), hip = csrc/conv1x1.hip (COT_CONV1X1=hip), matmul = batched GEMMs.
Reports forward and forward+backward time per call (HIP events, L2-cold rotation over 3 buffer sets) and the
fraction of the HBM roofline for the hip kernels' algorithmic bytes (fwd: (Ci+Co)*N*HW*2, bwd: 2x that + weights).

    python scripts/bench_conv1x1.py [--batch 80] [--iters 30]
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from torch import nn  # noqa: E402

from cotnet_amd import conv1x1 as c1  # noqa: E402

SHAPES = [  # (name, Ci, Co, H, split, bias)
    ("s1 conv1   256->64  @56", 256, 64, 56, 0, False),
    ("s1 embed0  128->32  @56", 128, 32, 56, 64, False),
    ("s1 embed3   32->72  @56", 32, 72, 56, 0, True),
    ("s1 conv1x1  64->64  @56", 64, 64, 56, 0, False),
    ("s1 conv3    64->256 @56", 64, 256, 56, 0, False),
    ("s2 conv1   512->128 @28", 512, 128, 28, 0, False),
    ("s2 embed0  256->64  @28", 256, 64, 28, 128, False),
    ("s2 conv3   128->512 @28", 128, 512, 28, 0, False),
    ("s3 conv1  1024->256 @14", 1024, 256, 14, 0, False),
    ("s3 embed0  512->128 @14", 512, 128, 14, 256, False),
    ("s3 conv3   256->1024@14", 256, 1024, 14, 0, False),
    ("s4 conv1  2048->512 @7 ", 2048, 512, 7, 0, False),
    ("s4 embed0 1024->256 @7 ", 1024, 256, 7, 512, False),
    ("s4 conv3   512->2048@7 ", 512, 2048, 7, 0, False),
]


def time_calls(fn, iters, warm=5):
    for _ in range(warm):
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
    ap.add_argument("--iters", type=int, default=30)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.backends.cudnn.benchmark = True
    N = args.batch
    print(f"{'shape':26s} {'mode':7s} {'fwd us':>9s} {'f+b us':>9s} {'fwd %HBM':>9s} {'bwd %HBM':>9s}")
    for name, Ci, Co, H, split, bias in SHAPES:
        conv = nn.Conv2d(Ci, Co, 1, bias=bias).to(dev).bfloat16()
        sets = []
        for _ in range(3):
            x = torch.randn(N, Ci, H, H, device=dev).bfloat16()
            xs = [x[:, :split].contiguous(), x[:, split:].contiguous()] if split else [x]
            xs = [t.requires_grad_(True) for t in xs]
            sets.append((xs, torch.randn(N, Co, H, H, device=dev).bfloat16()))
        fwd_bytes = (Ci + Co) * N * H * H * 2
        bwd_bytes = 2 * fwd_bytes + 2 * Ci * Co * 2
        for mode in ("", "hip", "matmul"):
            c1.MODE = mode
            it = [0]

            def fwd():
                xs, _ = sets[it[0] % 3]
                it[0] += 1
                with torch.no_grad():
                    c1.conv1x1(conv, *xs)

            def fwd_bwd():
                xs, gy = sets[it[0] % 3]
                it[0] += 1
                for t in xs:
                    t.grad = None
                conv.weight.grad = None
                c1.conv1x1(conv, *xs).backward(gy)

            tf = time_calls(fwd, args.iters)
            tfb = time_calls(fwd_bwd, args.iters)
            pf = fwd_bytes / (tf * 1e-6) / 8e12 * 100
            pb = bwd_bytes / (max(tfb - tf, 1e-3) * 1e-6) / 8e12 * 100
            print(f"{name:26s} {mode or 'module':7s} {tf:9.1f} {tfb:9.1f} {pf:9.1f} {pb:9.1f}", flush=True)
    c1.MODE = ""


if __name__ == "__main__":
    main()

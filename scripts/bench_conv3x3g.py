#!/usr/bin/env python
"""A/B of the grouped 3x3 convolution (CotLayer.key_embed) at the CoTNet-50 stage shapes, B=80, bf16:
module (MIOpen / CK) vs COT_CONV3X3=hip (csrc/conv3x3g.hip).  Forward and forward+backward time per call,
rotating over 3 buffer sets; % of the 8 TB/s HBM roofline for the algorithmic bytes (fwd: x + y; bwd: 2x that).

    python scripts/bench_conv3x3g.py [--batch 80] [--iters 30]
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from torch import nn  # noqa: E402

from cotnet_amd import conv3x3g as c3  # noqa: E402


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
    return e0.elapsed_time(e1) / iters * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=80)
    ap.add_argument("--iters", type=int, default=30)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.backends.cudnn.benchmark = True
    N = args.batch
    print(f"{'shape':22s} {'mode':7s} {'fwd us':>9s} {'f+b us':>9s} {'fwd %HBM':>9s} {'bwd %HBM':>9s}")
    for C, G, H in ((64, 4, 56), (128, 4, 28), (256, 4, 14), (512, 4, 7), (192, 8, 28)):
        conv = nn.Conv2d(C, C, 3, padding=1, groups=G, bias=False).to(dev).bfloat16()
        sets = [(torch.randn(N, C, H, H, device=dev).bfloat16().requires_grad_(True),
                 torch.randn(N, C, H, H, device=dev).bfloat16()) for _ in range(3)]
        fwd_bytes = 2 * C * N * H * H * 2
        for mode in ("", "hip"):
            c3.MODE = mode
            it = [0]

            def fwd():
                x, _ = sets[it[0] % 3]
                it[0] += 1
                with torch.no_grad():
                    c3.conv3x3(conv, x)

            def fwd_bwd():
                x, gy = sets[it[0] % 3]
                it[0] += 1
                x.grad = None
                conv.weight.grad = None
                c3.conv3x3(conv, x).backward(gy)

            tf, tfb = time_calls(fwd, args.iters), time_calls(fwd_bwd, args.iters)
            pf = fwd_bytes / (tf * 1e-6) / 8e12 * 100
            pb = 2 * fwd_bytes / (max(tfb - tf, 1e-3) * 1e-6) / 8e12 * 100
            print(f"C{C} g{G} {H}x{H}".ljust(22) + f" {mode or 'module':7s} {tf:9.1f} {tfb:9.1f} {pf:9.1f} {pb:9.1f}",
                  flush=True)
    c3.MODE = ""


if __name__ == "__main__":
    main()

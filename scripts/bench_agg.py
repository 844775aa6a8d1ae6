#!/usr/bin/env python
"""Micro-benchmark of the aggregation kernels on the CoTNet-50 CoT-layer geometries.

For each (C, H) stage shape at the per-GPU batch, dtype and layout: forward and fused backward, HIP-event timed on the
launch stream over `--iters` launches after warm-up.  `cold` rotates over enough buffer sets to exceed the 256 MiB
Infinity Cache (HBM-resident inputs); `hot` reuses one set (what a producer->consumer chain inside the model sees).
Algorithmic bytes: fwd e*(x + w + out) = 3.125*e per output element; fused bwd e*(gO + x + w + gX + gW) = 5.25*e.
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from cotnet_amd import _lib  # noqa: E402
from cotnet_amd.aggregation_zeropad import aggregation_zeropad  # noqa: E402

SHAPES = [(64, 56), (128, 28), (256, 14), (512, 7)]
PEAK = 8000.0


def make(N, C, HW, dtype, layout, dev):
    x = torch.randn(N, C, HW, HW, device=dev, dtype=dtype)
    w = torch.randn(N, 1, C // 8, 9, HW, HW, device=dev, dtype=dtype)
    if layout == "nhwc":
        x = x.contiguous(memory_format=torch.channels_last)
        w = w.permute(0, 4, 5, 1, 2, 3).contiguous().permute(0, 3, 4, 5, 1, 2)
    return x.requires_grad_(True), w.requires_grad_(True)


def time_ms(fn_list, iters):
    for f in fn_list[:min(len(fn_list), 3)]:
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(iters):
        fn_list[i % len(fn_list)]()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=80)
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--dtypes", default="bf16,fp32")
    ap.add_argument("--layouts", default="nchw,nhwc")
    ap.add_argument("--out", default="")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    rows = []
    for dname in args.dtypes.split(","):
        dtype = {"bf16": torch.bfloat16, "fp32": torch.float32, "fp16": torch.float16}[dname]
        e = torch.empty((), dtype=dtype).element_size()
        for layout in args.layouts.split(","):
            for C, HW in SHAPES:
                N = args.batch
                elems = N * C * HW * HW
                fwd_bytes = e * elems * 3.125
                bwd_bytes = e * elems * 5.25
                nsets = max(2, int(600e6 // fwd_bytes) + 1)
                sets = [make(N, C, HW, dtype, layout, dev) for _ in range(min(nsets, 12))]
                outs = [aggregation_zeropad(x, w, 3, 1, 1, 1) for x, w in sets]
                fk = _lib.last_kernel()
                gos = [torch.randn_like(o) for o in outs]

                def fwd_fn(i):
                    x, w = sets[i]
                    return lambda: aggregation_zeropad(x.detach(), w.detach(), 3, 1, 1, 1)

                def bwd_fn(i):
                    return lambda: torch.autograd.grad(outs[i], sets[i], gos[i], retain_graph=True)

                res = {}
                for tag, fns in (("fwd_cold", [fwd_fn(i) for i in range(len(sets))]), ("fwd_hot", [fwd_fn(0)]),
                                 ("bwd_cold", [bwd_fn(i) for i in range(len(sets))]), ("bwd_hot", [bwd_fn(0)])):
                    ms = time_ms(fns, args.iters)
                    nbytes = fwd_bytes if tag.startswith("fwd") else bwd_bytes
                    res[tag] = (ms, nbytes / (ms * 1e-3) / 1e9)
                bk = _lib.last_kernel()
                row = dict(dtype=dname, layout=layout, C=C, HW=HW, N=N, fwd_kernel=fk, bwd_kernel=bk,
                           **{f"{k}_us": round(v[0] * 1e3, 1) for k, v in res.items()},
                           **{f"{k}_GBs": round(v[1], 0) for k, v in res.items()},
                           **{f"{k}_frac": round(v[1] / PEAK, 3) for k, v in res.items()})
                rows.append(row)
                print(f"{dname:5s} {layout} C{C:<4d} {HW:>2d}x{HW:<2d}  fwd cold {res['fwd_cold'][0]*1e3:8.1f} us "
                      f"{res['fwd_cold'][1]:7.0f} GB/s ({res['fwd_cold'][1]/PEAK:5.1%})  hot {res['fwd_hot'][1]:7.0f} GB/s | "
                      f"bwd cold {res['bwd_cold'][0]*1e3:8.1f} us {res['bwd_cold'][1]:7.0f} GB/s "
                      f"({res['bwd_cold'][1]/PEAK:5.1%})  hot {res['bwd_hot'][1]:7.0f} GB/s   [{fk} | {bk}]", flush=True)
                del sets, outs, gos
                torch.cuda.empty_cache()
    if args.out:
        os.makedirs(os.path.dirname(args.out) or ".", exist_ok=True)
        json.dump(rows, open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()

#!/usr/bin/env python
"""Kernel-variant A/B of the aggregation kernels straight through the C ABI (no autograd / allocator in the loop).

Interleaved rounds over the variants (cot_set_tuning), HIP events on the launch stream, rotating buffer sets that
exceed the 256 MiB Infinity Cache ("cold") or one set ("hot").  Prints median-of-rounds per variant.
"""
import argparse
import ctypes
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from cotnet_amd import _lib  # noqa: E402

PEAK = 8000.0
SHAPES = [(64, 56), (128, 28), (256, 14), (512, 7)]


def P(t):
    return ctypes.c_void_p(t.data_ptr())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=80)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--dtypes", default="bf16,fp32")
    ap.add_argument("--shapes", default="0,1,2,3")
    ap.add_argument("--variants", default="v1,v2dpp,v2shfl")
    ap.add_argument("--out", default="")
    args = ap.parse_args()
    L = _lib.lib()
    dev = torch.device("cuda:0")
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    print("xchg probe mode:", L.cot_xchg_mode(), flush=True)
    VAR = {
        "v1": [(0, 1)], "v2dpp": [(0, 2), (3, 0)], "v2shfl": [(0, 2), (3, 1)],
        "v2dpp_fP4": [(0, 2), (3, 0), (1, 4)], "v2dpp_bP2": [(0, 2), (3, 0), (2, 2)],
        "v2dpp_fP2": [(0, 2), (3, 0), (1, 2)],
        "v2": [(0, 2), (1, 4)], "v2_fP8": [(0, 2), (1, 8)],
        "v3": [(0, 3), (1, 4), (4, 4)], "v3_fP8": [(0, 3), (1, 8), (4, 4)], "v3_jp2": [(0, 3), (1, 4), (4, 2)],
        "v3_jp8": [(0, 3), (1, 8), (4, 8)], "v3_bP2": [(0, 3), (1, 4), (2, 2), (4, 4)],
        "v3d": [(0, 3)], "v3d_nw8": [(0, 3), (5, 8)], "v3d_pad16": [(0, 3), (6, 16)], "v3d_pad32": [(0, 3), (6, 32)],
        "v3d_pad56": [(0, 3), (6, 56)], "v3d_nw8_pad32": [(0, 3), (5, 8), (6, 32)], "v3d_bP4": [(0, 3), (2, 4)],
        "v3d_bP4_pad32": [(0, 3), (2, 4), (6, 32)],
        "v3d_xcd": [(0, 3), (7, 1)], "v3d_split": [(0, 3), (8, 1)], "v3d_xcd_split": [(0, 3), (7, 1), (8, 1)],
        "v3d_jp8": [(0, 3), (4, 8)], "v3d_jp2": [(0, 3), (4, 2)],
        # automatic dispatch (key 0 = 0): bf16 fused backward on the packed dot-product kernel (agg_dot2.hip); keys 30 / 32 / 33 =
        # channel groups per LDS phase, waves per workgroup, SAFE operand masking; "lds" = the same dispatch with it off
        "lds": [(29, 0)], "dot2": [],
        # automatic dispatch with the forward's pixels per lane / waves per workgroup changed (round 6 re-check of the small planes)
        "fP2": [(1, 2)], "fP8": [(1, 8)], "nw8": [(5, 8)], "fP2_nw8": [(1, 2), (5, 8)], "fP1": [(1, 1)],
    }
    for jp in (2, 4, 8):
        for nw in (2, 4, 7):
            for safe in (0, 1):
                VAR[f"d2_jp{jp}_nw{nw}_s{safe}"] = [(30, jp), (32, nw), (33, safe)]
                VAR[f"d2_jp{jp}_nw{nw}_s{safe}_sb"] = [(30, jp), (32, nw), (33, safe), (34, 0)]
                VAR[f"d2_jp{jp}_nw{nw}_s{safe}_xcd"] = [(30, jp), (32, nw), (33, safe), (31, 1)]

    def set_variant(name):
        for k, v in ((0, 0), (1, 4), (2, 2), (3, -1), (4, 0), (5, 4), (6, 0), (7, -1), (8, 0), (29, 1), (30, 0), (31, -1), (32, 0), (33, 1), (34, 1)):
            L.cot_set_tuning(k, v)
        for k, v in VAR[name]:
            L.cot_set_tuning(k, v)

    rows = []
    for dname in args.dtypes.split(","):
        dtype = {"bf16": torch.bfloat16, "fp32": torch.float32}[dname]
        e = torch.empty((), dtype=dtype).element_size()
        code = _lib.dtype_code(dtype)
        for si in [int(s) for s in args.shapes.split(",")]:
            C, HW = SHAPES[si]
            N = args.batch
            elems = N * C * HW * HW
            g = _lib.AggGeom(N, C, HW, HW, 1, C // 8, 3, 3, 1, 1, 1, 1, 1, 1)
            nsets = min(12, max(2, int(700e6 // (e * elems * 5.25)) + 1))
            sets = []
            for _ in range(nsets):
                x = torch.randn(N, C, HW, HW, device=dev, dtype=dtype)
                w = torch.randn(N, 1, C // 8, 9, HW, HW, device=dev, dtype=dtype)
                go = torch.randn(N, C, HW, HW, device=dev, dtype=dtype)
                sets.append((x, w, go, torch.empty_like(x), torch.empty_like(x), torch.empty_like(w)))

            def fwd(i):
                x, w, go, out, gx, gw = sets[i]
                rc = L.cot_agg_forward(P(x), P(w), P(out), ctypes.byref(g), code, 0, stream)
                assert rc == 0, L.cot_last_error()

            def bwd(i):
                x, w, go, out, gx, gw = sets[i]
                rc = L.cot_agg_backward(P(go), P(x), P(w), P(gx), P(gw), ctypes.byref(g), code, 0, stream)
                assert rc == 0, L.cot_last_error()

            def timeit(fn, cold):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                fn(0)
                torch.cuda.synchronize()
                e0.record()
                for i in range(args.iters):
                    fn(i % nsets if cold else 0)
                e1.record()
                torch.cuda.synchronize()
                return e0.elapsed_time(e1) / args.iters * 1e3  # us

            res = {}
            names = args.variants.split(",")
            ref_out = None
            for rnd in range(args.rounds):
                for v in names:
                    set_variant(v)
                    for kind, fn, nb in (("fwd", fwd, 3.125), ("bwd", bwd, 5.25)):
                        for cold in (True, False):
                            us = timeit(fn, cold)
                            res.setdefault((v, kind, cold), []).append(us)
                    if rnd == 0:  # cross-check variants against each other on set 0
                        fwd(0), bwd(0)
                        torch.cuda.synchronize()
                        cur = [t.float().clone() for t in sets[0][3:]]
                        kern = L.cot_last_kernel().decode()
                        if ref_out is None:
                            ref_out = cur
                        else:
                            errs = [float((a - b).abs().max()) for a, b in zip(cur, ref_out)]
                            print(f"   {v} vs {names[0]} max|diff| out/gx/gw = {errs}  [{kern}]", flush=True)
            for v in names:
                row = dict(dtype=dname, C=C, HW=HW, N=N, variant=v)
                msg = f"{dname:5s} C{C:<4d}{HW:>3d}x{HW:<3d} {v:10s}"
                for kind, nb in (("fwd", 3.125), ("bwd", 5.25)):
                    for cold in (True, False):
                        us = sorted(res[(v, kind, cold)])[len(res[(v, kind, cold)]) // 2]
                        gbs = e * elems * nb / (us * 1e-6) / 1e9
                        tag = f"{kind}_{'cold' if cold else 'hot'}"
                        row[tag + "_us"], row[tag + "_GBs"], row[tag + "_frac"] = round(us, 1), round(gbs), round(gbs / PEAK, 3)
                        msg += f" | {tag} {us:7.1f}us {gbs:6.0f}GB/s {gbs / PEAK:5.1%}"
                rows.append(row)
                print(msg, flush=True)
            del sets
            torch.cuda.empty_cache()
    for k, v in ((0, 0), (1, 4), (2, 2), (3, -1), (4, 0), (5, 4), (6, 0), (7, -1), (8, 0), (29, 1), (30, 0), (31, -1), (32, 0), (33, 1), (34, 1)):
        L.cot_set_tuning(k, v)
    if args.out:
        json.dump(rows, open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()

#!/usr/bin/env python
"""Device time of the 1x1-convolution entry points straight through the C ABI (no autograd, no allocator in the loop):
forward / data gradient / weight gradient at every 1x1 shape of CoTNet-50 (B = 80, bf16), first-generation kernels
(cot_set_tuning(15, 0)) against the LDS-pipelined ones (15, 1).  HIP events around `iters` back-to-back launches on
rotating buffer sets (cold) -- per-launch time, fraction of the 8 TB/s HBM roofline for the algorithmic bytes
2*N*HW*(Ci+Co) (+ weights), and MFMA TFLOP/s.

    python scripts/bench_conv_abi.py [--batch 80] [--iters 20] [--json out.json]
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

SHAPES = [  # (name, Ci, Co, H, split, bias, per-step count fwd)
    ("s1 conv1   256->64  @56", 256, 64, 56, 0, False),
    ("s1 embed0  128->32  @56", 128, 32, 56, 64, False),
    ("s1 embed3   32->72  @56", 32, 72, 56, 0, True),
    ("s1 conv1x1  64->64  @56", 64, 64, 56, 0, False),
    ("s1 conv3    64->256 @56", 64, 256, 56, 0, False),
    ("s2 conv1   512->128 @28", 512, 128, 28, 0, False),
    ("s2 embed0  256->64  @28", 256, 64, 28, 128, False),
    ("s2 embed3   64->144 @28", 64, 144, 28, 0, True),
    ("s2 conv1x1 128->128 @28", 128, 128, 28, 0, False),
    ("s2 conv3   128->512 @28", 128, 512, 28, 0, False),
    ("s3 conv1  1024->256 @14", 1024, 256, 14, 0, False),
    ("s3 embed0  512->128 @14", 512, 128, 14, 256, False),
    ("s3 embed3  128->288 @14", 128, 288, 14, 0, True),
    ("s3 conv1x1 256->256 @14", 256, 256, 14, 0, False),
    ("s3 conv3   256->1024@14", 256, 1024, 14, 0, False),
    ("s4 conv1  2048->512 @7 ", 2048, 512, 7, 0, False),
    ("s4 embed0 1024->256 @7 ", 1024, 256, 7, 512, False),
    ("s4 embed3  256->576 @7 ", 256, 576, 7, 0, True),
    ("s4 conv1x1 512->512 @7 ", 512, 512, 7, 0, False),
    ("s4 conv3   512->2048@7 ", 512, 2048, 7, 0, False),
    ("se  fc1    512->256 x80", 512, 256, None, 0, True),   # the se branch: one "image" whose 80 pixels are the batch
    ("se  fc2    256->1024x80", 256, 1024, None, 0, True),
]


def P(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=80)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--json", default="")
    ap.add_argument("--modes", default="0,1")
    ap.add_argument("--only", default="", help="substring filter on the shape name (e.g. 's4 conv1')")
    ap.add_argument("--tune", default="", help="KEY=VALUE[,KEY=VALUE...] for cot_set_tuning (A/B)")
    args = ap.parse_args()
    L = _lib.lib()
    for kv in filter(None, args.tune.split(",")):
        key, value = kv.split("=")
        assert L.cot_set_tuning(int(key), int(value)) == 0
    dev = torch.device("cuda:0")
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    BF = _lib.COT_BF16
    rows = []
    print(f"{'shape':26s} {'gen':3s} {'fwd us':>8s} {'%HBM':>6s} {'TF/s':>6s} | {'dgrad us':>8s} {'%HBM':>6s} | {'wgrad us':>8s} {'%HBM':>6s}")
    for name, Ci, Co, H, split, bias in SHAPES:
        if args.only and args.only not in name:
            continue
        N, HW = (args.batch, H * H) if H else (1, args.batch)
        nset = max(2, min(6, int(300e6 // ((Ci + Co) * N * HW * 2)) + 1))  # rotate through > 256 MiB where it fits
        sets = []
        for _ in range(nset):
            x = torch.randn(N, Ci, HW, device=dev).bfloat16()
            xs = (x[:, :split].contiguous(), x[:, split:].contiguous()) if split else (x, None)
            sets.append((xs, torch.randn(N, Co, HW, device=dev).bfloat16(), torch.empty(N, Co, HW, device=dev).bfloat16(),
                         (torch.empty(N, split, HW, device=dev).bfloat16(), torch.empty(N, Ci - split, HW, device=dev).bfloat16())
                         if split else (torch.empty(N, Ci, HW, device=dev).bfloat16(), None)))
        w = (torch.randn(Co, Ci, device=dev) / Ci ** 0.5).bfloat16()
        b = torch.randn(Co, device=dev).bfloat16() if bias else None
        gw, gb = torch.empty_like(w), (torch.empty_like(b) if bias else None)
        ws = torch.empty(int(L.cot_conv1x1_workspace(N, Ci, Co, HW, 1 if bias else 0)), dtype=torch.uint8, device=dev)
        c1 = split if split else Ci
        act_bytes = 2 * N * HW * (Ci + Co)
        flops = 2.0 * N * HW * Ci * Co

        def timed(fn):
            for i in range(3):
                fn(i)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(args.iters):
                fn(i)
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) / args.iters * 1e3

        def fwd(i):
            (x1, x2), gy, y, gx = sets[i % nset]
            rc = L.cot_conv1x1_forward(P(x1), P(x2), c1, P(w), P(b), P(y), N, Ci, Co, HW, BF, st)
            assert rc == 0, L.cot_last_error()

        def dgrad(i):
            (x1, x2), gy, y, (g1, g2) = sets[i % nset]
            rc = L.cot_conv1x1_backward_data(P(gy), P(w), P(g1), P(g2), c1, 0, P(ws), N, Ci, Co, HW, BF, st)
            assert rc == 0, L.cot_last_error()

        def wgrad(i):
            (x1, x2), gy, y, gx = sets[i % nset]
            rc = L.cot_conv1x1_backward_weight(P(gy), P(x1), P(x2), c1, P(gw), P(gb), P(ws), N, Ci, Co, HW, BF, st)
            assert rc == 0, L.cot_last_error()

        outs = {}
        for gen in [int(m) for m in args.modes.split(",")]:
            assert L.cot_set_tuning(15, gen) == 0
            tf, td, tw = timed(fwd), timed(dgrad), timed(wgrad)
            outs[gen] = sets[0][2].float().clone()
            pf, pd, pw = (100 * act_bytes / (t * 1e-6) / 8e12 for t in (tf, td, tw))
            print(f"{name:26s} {gen:3d} {tf:8.1f} {pf:6.1f} {flops / (tf * 1e-6) / 1e12:6.0f} | {td:8.1f} {pd:6.1f} | {tw:8.1f} {pw:6.1f}",
                  flush=True)
            rows.append({"shape": name.strip(), "gen": gen, "fwd_us": round(tf, 2), "dgrad_us": round(td, 2),
                         "wgrad_us": round(tw, 2), "fwd_frac_hbm": round(pf / 100, 4), "dgrad_frac_hbm": round(pd / 100, 4),
                         "wgrad_frac_hbm": round(pw / 100, 4), "fwd_tflops": round(flops / (tf * 1e-6) / 1e12, 1)})
        if len(outs) == 2:  # both generations on the last buffer set: same result up to summation order
            fwd(0)
            torch.cuda.synchronize()
            d = (outs[0] - outs[1]).abs().max().item()
            if d > 0.05 * outs[0].abs().max().item() + 1e-3:
                print(f"   !! generations disagree: max|diff| {d:.3e}")
        L.cot_set_tuning(15, 1)
    # ---- grouped 3x3 key-embed convolutions (CoTNet-50: groups 4)
    print(f"\n{'grouped 3x3':26s} {'gen':3s} {'fwd us':>8s} {'%HBM':>6s} {'TF/s':>6s} | {'dgrad us':>8s} {'%HBM':>6s} | {'wgrad us':>8s} {'%HBM':>6s}")
    for C, H in ((64, 56), (128, 28), (256, 14), (512, 7)):
        if args.only and args.only not in f"C{C} g4":
            continue
        N, G = args.batch, 4
        nset = max(2, min(6, int(300e6 // (2 * C * N * H * H * 2)) + 1))
        sets = [(torch.randn(N, C, H, H, device=dev).bfloat16(), torch.randn(N, C, H, H, device=dev).bfloat16(),
                 torch.empty(N, C, H, H, device=dev).bfloat16()) for _ in range(nset)]
        w3 = (torch.randn(C, C // G, 3, 3, device=dev) / (9 * C // G) ** 0.5).bfloat16()
        gw3 = torch.empty_like(w3)
        masks = torch.empty(int(L.cot_conv3x3g_masks_bytes(H, H)), dtype=torch.uint8, device=dev)
        assert L.cot_conv3x3g_masks(P(masks), H, H, st) == 0
        ws3 = torch.empty(int(L.cot_conv3x3g_workspace(N, C, C, G, H, H)), dtype=torch.uint8, device=dev)
        act_bytes = 2 * N * H * H * 2 * C
        flops = 2.0 * 9 * (C // G) * C * N * H * H

        def timed3(fn):
            for i in range(3):
                fn(i)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(args.iters):
                fn(i)
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) / args.iters * 1e3

        def f3(i):
            x, gy, y = sets[i % nset]
            assert L.cot_conv3x3g_forward(P(x), P(w3), P(y), P(masks), P(ws3), N, C, C, G, H, H, BF, st) == 0, L.cot_last_error()

        def d3(i):
            x, gy, y = sets[i % nset]
            assert L.cot_conv3x3g_backward_data(P(gy), P(w3), P(y), 0, P(masks), P(ws3), N, C, C, G, H, H, BF, st) == 0

        def w3g(i):
            x, gy, y = sets[i % nset]
            assert L.cot_conv3x3g_backward_weight(P(gy), P(x), P(gw3), P(masks), P(ws3), N, C, C, G, H, H, BF, st) == 0

        for gen in [int(m) for m in args.modes.split(",")]:
            assert L.cot_set_tuning(15, gen) == 0
            tf, td, tw = timed3(f3), timed3(d3), timed3(w3g)
            pf, pd, pw = (100 * act_bytes / (t * 1e-6) / 8e12 for t in (tf, td, tw))
            name = f"C{C} g4 {H}x{H}"
            print(f"{name:26s} {gen:3d} {tf:8.1f} {pf:6.1f} {flops / (tf * 1e-6) / 1e12:6.0f} | {td:8.1f} {pd:6.1f} | {tw:8.1f} {pw:6.1f}", flush=True)
            rows.append({"shape": "conv3x3g " + name, "gen": gen, "fwd_us": round(tf, 2), "dgrad_us": round(td, 2),
                         "wgrad_us": round(tw, 2), "fwd_frac_hbm": round(pf / 100, 4), "dgrad_frac_hbm": round(pd / 100, 4),
                         "wgrad_frac_hbm": round(pw / 100, 4), "fwd_tflops": round(flops / (tf * 1e-6) / 1e12, 1)})
        L.cot_set_tuning(15, 1)
    if args.json:
        json.dump(rows, open(args.json, "w"), indent=1)


if __name__ == "__main__":
    main()

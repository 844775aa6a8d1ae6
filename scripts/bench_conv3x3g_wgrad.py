#!/usr/bin/env python
"""Grouped 3x3 weight gradient through the C ABI at CoTNet-50's key_embed shapes (B = 80, bf16): the per-wave kernel
(cot_conv3x3g_backward_weight, csrc/conv3x3g.hip) against the LDS-staged one (cot_conv3x3g_backward_weight_guarded with
x inside a larger allocation, csrc/conv_wgrad2.hip TAPS form).  Per launch (incl. the reduce kernel): time, % of the 8 TB/s
HBM roofline for the algorithmic bytes (x + dY read once, dW written once), relative error against torch's conv2d weight gradient
in fp32 on the same bf16 operands.

    python scripts/bench_conv3x3g_wgrad.py [--batch 80] [--iters 20] [--tune 25=...]"""
import argparse
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from cotnet_amd import _lib  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=80)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--tune", default="")
    ap.add_argument("--no-check", action="store_true")
    args = ap.parse_args()
    L = _lib.lib()
    for kv in filter(None, args.tune.split(",")):
        k, v = kv.split("=")
        assert L.cot_set_tuning(int(k), int(v)) == 0
    dev = torch.device("cuda:0")
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    P = lambda t: ctypes.c_void_p(t.data_ptr())
    N = args.batch
    print(f"{'shape':18s} {'kernel':10s} {'us':>8s} {'%HBM':>6s} {'TF/s':>6s} {'rel err':>9s}")
    for C, G, H in ((64, 4, 56), (128, 4, 28), (256, 4, 14), (512, 4, 7)):
        HW, W = H * H, H
        lead = (W + 1 + 7) // 8 * 8
        sets = []
        for _ in range(3):
            flat = torch.randn(N * C * HW + 2 * lead, device=dev).bfloat16()
            sets.append((flat[lead:lead + N * C * HW].view(N, C, H, H), torch.randn(N, C, H, H, device=dev).bfloat16()))
        masks = torch.empty(int(L.cot_conv3x3g_masks_bytes(H, H)), dtype=torch.uint8, device=dev)
        assert L.cot_conv3x3g_masks(P(masks), H, H, st) == 0
        ws = torch.empty(int(L.cot_conv3x3g_workspace(N, C, C, G, H, H)), dtype=torch.uint8, device=dev)
        gw = torch.empty(C, C // G, 3, 3, device=dev).bfloat16()
        ref = None
        if not args.no_check:
            x, gy = sets[0]
            wf = torch.zeros(C, C // G, 3, 3, device=dev, requires_grad=True)
            torch.nn.functional.conv2d(x.float(), wf, None, 1, 1, 1, G).backward(gy.float())
            ref = wf.grad
        nbytes = 2 * N * C * HW * 2 + C * (C // G) * 9 * 2
        flops = 2.0 * N * HW * C * (C // G) * 9
        for name, guard in (("per-wave", 0), ("lds-staged", lead)):
            it = [0]

            def run():
                x, gy = sets[it[0] % 3]
                it[0] += 1
                rc = L.cot_conv3x3g_backward_weight_guarded(P(gy), P(x), P(gw), P(masks), P(ws), N, C, C, G, H, H, _lib.COT_BF16, guard, st)
                assert rc == 0, L.cot_last_error()
            err = float("nan")
            if ref is not None:
                it[0] = 0
                gw.fill_(float("nan"))
                run()
                torch.cuda.synchronize()
                err = ((gw.float() - ref).abs().max() / ref.abs().max()).item()
            for _ in range(3):
                run()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(args.iters):
                run()
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) / args.iters * 1e3
            print(f"C{C} g{G} {H}x{H}".ljust(18) + f" {name:10s} {us:8.1f} {nbytes / (us * 1e-6) / 8e12 * 100:6.1f} {flops / (us * 1e-6) * 1e-12:6.0f} {err:9.2e}"
                  "", flush=True)


if __name__ == "__main__":
    main()

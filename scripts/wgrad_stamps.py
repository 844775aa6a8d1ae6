#!/usr/bin/env python
"""Where does a weight-gradient launch spend its time?  The third-generation kernel (csrc/conv_wgrad2.hip) stamps s_memtime
per workgroup at: 0 start, 1 set-up done, 2 first K step done (first data arrived), 3 loop done, 4 stores issued, 5 stores
retired.  Printed per shape: launch wall time (HIP events, main kernel + reduce), and for each interval the median / max over
the workgroups in microseconds (ticks calibrated against the wall time of the whole launch).

    python scripts/wgrad_stamps.py [--tune 25=...]"""
import argparse
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from cotnet_amd import _lib  # noqa: E402

SHAPES = [("s1 conv1 256->64 @56", 256, 64, 56), ("s1 conv3 64->256 @56", 64, 256, 56), ("s2 conv1 512->128 @28", 512, 128, 28),
          ("s3 conv1 1024->256 @14", 1024, 256, 14), ("s3 conv1x1 256->256 @14", 256, 256, 14), ("s4 conv1 2048->512 @7", 2048, 512, 7),
          ("s4 conv3 512->2048 @7", 512, 2048, 7)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tune", default="")
    ap.add_argument("--batch", type=int, default=80)
    args = ap.parse_args()
    L = _lib.lib()
    L_raw = ctypes.CDLL(_lib.LIB_PATH)
    L_raw.cot_debug_stamps.argtypes = [ctypes.c_void_p]
    for kv in filter(None, args.tune.split(",")):
        k, v = kv.split("=")
        assert L.cot_set_tuning(int(k), int(v)) == 0
    dev = torch.device("cuda:0")
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    P = lambda t: ctypes.c_void_p(t.data_ptr())
    N = args.batch
    print(f"{'shape':26s} {'wall us':>8s} {'WGs':>5s} | setup  1st-data   loop  stores  drain | (median / max us per workgroup; start skew max)")
    for name, Ci, Co, H in SHAPES:
        HW = H * H
        x = torch.randn(N, Ci, HW, device=dev).bfloat16()
        gy = torch.randn(N, Co, HW, device=dev).bfloat16()
        gw = torch.empty(Co, Ci, device=dev).bfloat16()
        ws = torch.empty(int(L.cot_conv1x1_workspace(N, Ci, Co, HW, 0)), dtype=torch.uint8, device=dev)
        stamps = torch.zeros(4096 * 8, dtype=torch.int64, device=dev)

        def run():
            rc = L.cot_conv1x1_backward_weight(P(gy), P(x), None, Ci, P(gw), None, P(ws), N, Ci, Co, HW, _lib.COT_BF16, st)
            assert rc == 0, L.cot_last_error()
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            run()
        e1.record()
        torch.cuda.synchronize()
        wall = e0.elapsed_time(e1) / 10 * 1e3
        L_raw.cot_debug_stamps(ctypes.c_void_p(stamps.data_ptr()))
        run()
        torch.cuda.synchronize()
        L_raw.cot_debug_stamps(None)
        s = stamps.view(-1, 8).cpu()
        s = s[s[:, 0] != 0]
        t = s[:, :6].double()
        t0 = t[:, 0].min()
        span = (t[:, 5].max() - t0).item()
        # ticks per microsecond: s_memtime runs at a fixed 100 MHz on gfx9 family parts
        tpu = 1.0
        d = (t[:, 1:6] - t[:, 0:5]) / tpu
        med, mx = d.median(0).values.tolist(), d.max(0).values.tolist()
        skew = ((t[:, 0] - t0) / tpu).max().item()
        xcc = torch.bincount(s[:, 7].clamp(0, 15), minlength=8).tolist()
        print(f"{name:26s} {wall:8.1f} {len(s):5d} | " + "  ".join(f"{a:5.1f}/{b:5.1f}" for a, b in zip(med, mx)) + "  (ticks)")
        ph = stamps.view(-1, 8)[2048:2056, :6].cpu().double()
        if (ph[:, 0] != 0).all():
            d = ph[:, 1:] - ph[:, :-1]
            nxt = ph[1:, 0] - ph[:-1, 5]
            print("      step phases (ticks, wave 0 of workgroup 0, steps 8..15): wait-copies | barrier | [prefetching form: issue reads | issue copies; else: issue copies | fragments] | issue MFMAs ; to next top")
            for i in range(8):
                print("      " + "  ".join(f"{int(v):6d}" for v in d[i].tolist()) + (f"  ; {int(nxt[i]):6d}" if i < 7 else ""))


if __name__ == "__main__":
    main()

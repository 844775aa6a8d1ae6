#!/usr/bin/env python
"""aggregation_zeropad_mix at the op-level shape BASELINE config 5 / SURVEY 8(d) name -- (B = 64, C = 256, 20 x 20, wC = 32,
heads = 1) -- straight through the C ABI (no autograd / allocator in the loop), HIP events on the launch stream.

Algorithmic bytes (what `GB/s` and `frac` are computed on; e = element size, HW = 400, per image):
    forward          e * HW * (C + 34 wC + 2 C)            x + w1 + w2 read, out (both kernel halves) written
    input backward   e * HW * (2 C + 34 wC + C)            gout + w1 + w2 read, gx written
    weight backward  e * HW * (2 C + C + 34 wC)            gout + x read, gw1 + gw2 written
Buffer sets rotate so that the working set exceeds the 256 MiB Infinity Cache ("cold"); `hot` re-uses one set.

    python scripts/bench_aggmix_abi.py [--batch 64] [--dtypes bf16,fp32] [--lanes 256,128,512] [--generic]
"""
import argparse
import ctypes
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
import torch  # noqa: E402

from cotnet_amd import _lib  # noqa: E402

PEAK_GBS = 8000.0
C, WC, H, W, HEADS = 256, 32, 20, 20, 1


def P(t):
    return ctypes.c_void_p(t.data_ptr())


def algorithmic_bytes(N, esize):
    hw = H * W
    return {"forward": esize * N * hw * (C + 34 * WC + 2 * C),
            "backward_input": esize * N * hw * (2 * C + 34 * WC + C),
            "backward_weight": esize * N * hw * (2 * C + C + 34 * WC)}


def measure(batch=64, dtype=torch.bfloat16, iters=20, rounds=5, cold=True, tuning=()):
    """-> {direction: {"us": median over rounds of the mean launch time, "GBps", "frac", "kernel"}}"""
    L = _lib.lib()
    dev = torch.device("cuda:0")
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    esize = torch.empty(0, dtype=dtype).element_size()
    per_set = esize * batch * H * W * (C + 34 * WC + 2 * C) * 2  # inputs + outputs of one set, all three directions
    nsets = max(1, min(8, -(-(320 << 20) // per_set))) if cold else 1
    g = torch.Generator(device="cpu").manual_seed(0)
    sets = []
    for _ in range(nsets):
        mk = lambda *s: torch.randn(*s, generator=g).to(dtype).to(dev)
        x, w1, w2 = mk(batch, C, H, W), mk(batch, HEADS, WC, 9, H, W), mk(batch, HEADS, WC, 25, H, W)
        gout = mk(batch, 2 * HEADS * C, H, W)
        sets.append((x, w1, w2, gout, torch.empty_like(gout), torch.empty_like(x), torch.empty_like(w1), torch.empty_like(w2)))
    geo = _lib.AggGeom(batch, C, H, W, HEADS, WC, 3, 3, 1, 1, 1, 1, 1, 1)
    dt = _lib.dtype_code(dtype)
    for k, v in tuning:
        L.cot_set_tuning(k, v)
    try:
        calls = {
            "forward": lambda s: L.cot_aggmix_forward(P(s[0]), P(s[1]), P(s[2]), P(s[4]), ctypes.byref(geo), 2, 2, dt, stream),
            "backward_input": lambda s: L.cot_aggmix_backward_input(P(s[3]), P(s[1]), P(s[2]), P(s[5]), ctypes.byref(geo), 2, 2, 0, dt, stream),
            "backward_weight": lambda s: L.cot_aggmix_backward_weight(P(s[3]), P(s[0]), P(s[6]), P(s[7]), ctypes.byref(geo), 2, 2, dt, stream),
        }
        alg = algorithmic_bytes(batch, esize)
        out = {}
        for name, call in calls.items():
            for s in sets:  # warm-up (first-touch, code load)
                assert call(s) == 0, L.cot_last_error()
            kernel = _lib.last_kernel()
            torch.cuda.synchronize()
            times = []
            for _ in range(rounds):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for i in range(iters):
                    call(sets[i % nsets])
                e1.record()
                e1.synchronize()
                times.append(e0.elapsed_time(e1) * 1e3 / iters)
            us = sorted(times)[len(times) // 2]
            gbs = alg[name] / us / 1e3
            out[name] = {"us": round(us, 2), "GBps": round(gbs, 1), "frac": round(gbs / PEAK_GBS, 4), "kernel": kernel,
                         "algorithmic_bytes": alg[name]}
        return out
    finally:
        for k, _ in tuning:
            L.cot_set_tuning(k, {51: 0, 52: 256, 53: 0}.get(k, 0))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--dtypes", default="bf16,fp32")
    ap.add_argument("--lanes", default="256")
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--generic", action="store_true", help="also time the one-lane-per-element kernels (cot_set_tuning(51, 1))")
    ap.add_argument("--hot", action="store_true")
    ap.add_argument("--ppl", default="0", help="pixels per lane to try (cot_set_tuning(53, .)); 0 = the planner's choice")
    ap.add_argument("--out", default="")
    args = ap.parse_args()
    res = {}
    for dn in args.dtypes.split(","):
        dtype = {"bf16": torch.bfloat16, "fp32": torch.float32, "fp16": torch.float16}[dn]
        variants = [(f"tile_lanes{l}" + (f"_P{pp}" if pp else ""), ((52, int(l)), (53, int(pp))))
                    for l in args.lanes.split(",") for pp in args.ppl.split(",")]
        if args.generic:
            variants.append(("generic", ((51, 1),)))
        for vn, tuning in variants:
            r = measure(args.batch, dtype, args.iters, args.rounds, not args.hot, tuning)
            res[f"{dn}|{vn}"] = r
            for d, v in r.items():
                print(f"{dn:5s} {vn:16s} {d:16s} {v['us']:8.2f} us  {v['GBps']:7.1f} GB/s  {v['frac']:.3f} of 8 TB/s  [{v['kernel']}]", flush=True)
    if args.out:
        with open(args.out, "w") as f:
            json.dump({"shape": {"N": args.batch, "C": C, "wC": WC, "H": H, "W": W, "heads": HEADS}, "results": res}, f, indent=1)


if __name__ == "__main__":
    main()

#!/usr/bin/env python
"""tests/test_fuzz_nodes_gpu.py's loop for many seeds: python scripts/fuzz_nodes_gpu.py [--seeds 30] [--count 20] [--first 8000] [--seconds 600]"""
import argparse
import os
import random
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402

from tests.test_fuzz_nodes_gpu import run_case  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", type=int, default=30)
    ap.add_argument("--count", type=int, default=20)
    ap.add_argument("--first", type=int, default=8000)
    ap.add_argument("--seconds", type=float, default=0)
    a = ap.parse_args()
    t0, total, bad, nodes, collected = time.time(), 0, 0, {}, []
    for seed in range(a.first, a.first + a.seeds):
        if a.seconds and time.time() - t0 > a.seconds:
            break
        rng = random.Random(seed)
        torch.manual_seed(seed)
        for _ in range(a.count):
            try:
                ok, desc = run_case(rng, collected)
            except Exception as e:  # noqa: BLE001  (a launch error, an exception in the host code: report and go on)
                ok, desc = False, ("EXCEPTION", repr(e)[:400])
            total += 1
            nodes[desc[-1] if ok else "failed"] = nodes.get(desc[-1] if ok else "failed", 0) + 1
            if not ok:
                bad += 1
                print("FAIL", seed, desc, flush=True)
    print(f"{total} stages in {time.time() - t0:.0f} s: {bad} failures")
    # both tails of err(candidate) / err(baseline) on the input gradient: a defect of the candidate shows as a one-sided tail, rounding noise
    # (small populations) as a two-sided one
    for name, sel in (("N <= 3", lambda d: d[5] <= 3), ("N > 3", lambda d: d[5] > 3)):
        r = sorted(rep["gx"][0] / max(rep["gx"][1], 1e-9) for d, rep in collected if sel(d))
        if r:
            q = lambda f: r[min(len(r) - 1, int(f * len(r)))]  # noqa: E731
            print(f"gx error ratio candidate / baseline, {name}: {len(r)} stages, median {q(0.5):.2f}, 5 % {q(0.05):.2f}, 95 % {q(0.95):.2f}, min {r[0]:.2f}, "
                  f"max {r[-1]:.2f}, above 2: {sum(v > 2 for v in r)}, below 1/2: {sum(v < 0.5 for v in r)}")
    print("first node of the stage's output:", dict(sorted(nodes.items())))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())

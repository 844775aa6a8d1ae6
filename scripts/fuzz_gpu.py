#!/usr/bin/env python
"""tests/test_fuzz_gpu.py's loop for many seeds on the device: python scripts/fuzz_gpu.py [--seeds 40] [--count 100] [--first 7000]
Prints the failing case descriptions (none expected) and the number of cases per kind."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from tests.test_fuzz_gpu import device_fuzz, run_cases  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", type=int, default=40)
    ap.add_argument("--count", type=int, default=100)
    ap.add_argument("--first", type=int, default=7000)
    ap.add_argument("--seconds", type=float, default=0, help="stop starting new seeds after this many seconds (0: no limit)")
    ap.add_argument("--scale", type=int, default=1, help="plane sizes of the convolution cases times this (tests.test_fuzz_emulated.SCALE)")
    a = ap.parse_args()
    t0, total, bad, kinds = time.time(), 0, [], {}
    with device_fuzz() as tfe:
        tfe.SCALE = a.scale
        for seed in range(a.first, a.first + a.seeds):
            if a.seconds and time.time() - t0 > a.seconds:
                break
            try:
                failures, k = run_cases(tfe, seed, a.count)
            except Exception as e:  # an assertion inside a case (a non-zero return code): report the seed and go on
                failures, k = [("EXCEPTION", seed, repr(e)[:300])], {}
            total += a.count
            for n, c in k.items():
                kinds[n] = kinds.get(n, 0) + c
            for f in failures:
                bad.append((seed, f))
                print("FAIL", seed, f, flush=True)
        tfe.SCALE = 1
    print(f"{total} cases over {seed - a.first + 1} seeds in {time.time() - t0:.0f} s: {len(bad)} failures")
    print("cases per kind:", dict(sorted(kinds.items())))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())

#!/usr/bin/env python
"""replay the seeds of scripts/fuzz_gpu.py that failed: python scripts/fuzz_gpu_one.py 7064 7075 ...  (prints the failing cases with the assertion's line)"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from tests.test_fuzz_gpu import device_fuzz, run_cases  # noqa: E402

with device_fuzz() as tfe:
    for seed in map(int, sys.argv[1:]):
        failures, _ = run_cases(tfe, seed, 100)
        for f in failures:
            print("FAIL", seed, f, flush=True)

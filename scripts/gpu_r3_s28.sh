#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp; O=gpurun_out
timeout 600 python -m pytest tests/test_conv3x3g_gpu.py -m gpu -x -q 2>&1 | tail -5 > $O/r3s28_tests.log
COT_NO_PROBE_CACHE=1 timeout 400 python bench.py --kernels new --steps 30 --warmup 10 > $O/r3s28_bench_a.json 2> $O/r3s28_bench_a.err
COT_NO_PROBE_CACHE=1 timeout 400 python bench.py --kernels new --steps 30 --warmup 10 > $O/r3s28_bench_b.json 2> $O/r3s28_bench_b.err
timeout 400 python bench.py > $O/r3s28_bench_default.json 2> $O/r3s28_bench_default.err

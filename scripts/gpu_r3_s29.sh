#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp; O=gpurun_out
timeout 300 python scripts/bench_pool.py > $O/r3s29_pool.log 2>&1
timeout 600 python -m pytest tests/test_pool_gpu.py tests/test_conv3x3g_gpu.py -m gpu -x -q 2>&1 | tail -3 >> $O/r3s29_pool.log
COT_NO_PROBE_CACHE=1 timeout 400 python bench.py --kernels new --steps 30 --warmup 10 > $O/r3s29_bench_a.json 2> $O/r3s29_bench_a.err
COT_NO_PROBE_CACHE=1 timeout 400 python bench.py --kernels new --steps 30 --warmup 10 --tune 27=0 > $O/r3s29_bench_b.json 2> $O/r3s29_bench_b.err

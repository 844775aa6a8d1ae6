#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp; O=gpurun_out
( time python bench.py > $O/r3s31_bench_default.json 2> $O/r3s31_bench_default.err ) 2> $O/r3s31_bench_default.time
timeout 900 python -m pytest tests/test_pool_gpu.py tests/test_fused_layer_gpu.py tests/test_layers_gpu.py -m gpu -x -q 2>&1 | tail -3 > $O/r3s31_tests.log

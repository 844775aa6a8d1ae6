#!/bin/bash
set -x
mkdir -p gpurun_out
timeout 100 python -m pytest tests/test_fused_bn_gpu.py tests/test_flat_sgd_gpu.py tests/test_layers_gpu.py -m gpu -q -x > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
tail -4 gpurun_out/pytest_gpu.log | cut -c1-300
timeout 60 python __graft_entry__.py smoke 2>&1 | tail -2
timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/bench_final2.json 2> gpurun_out/bench_final2.err; cut -c1-700 gpurun_out/bench_final2.json; tail -2 gpurun_out/bench_final2.err

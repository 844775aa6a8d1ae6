#!/bin/bash
# Round 2, session D: 8-wave workgroups for the LDS 1x1 kernels (A/B against 4-wave: COT_TUNING=17=1)
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
O=gpurun_out
timeout 300 python -m pytest tests/test_conv1x1_gpu.py tests/test_fused_layer_gpu.py -m gpu -q --timeout 240 -rfE -p no:cacheprovider > $O/r2d_pytest.log 2>&1; echo "pytest rc=$?" >> $O/r2d_pytest.log
grep -E "passed|failed|^FAILED|^ERROR|rc=" $O/r2d_pytest.log | cut -c1-300 | tail -20
timeout 300 python scripts/bench_conv_abi.py --iters 20 --modes 1 --json $O/r2d_conv_abi_w8.json > $O/r2d_conv_abi_w8.log 2>&1; tail -23 $O/r2d_conv_abi_w8.log | cut -c1-100
COT_TUNING=17=1 timeout 300 python scripts/bench_conv_abi.py --iters 20 --modes 1 > $O/r2d_conv_abi_w4.log 2>&1; tail -23 $O/r2d_conv_abi_w4.log | cut -c1-100
B="timeout 420 python bench.py --steps 20 --warmup 8 --no-cpu-baseline"
$B --kernels new > $O/r2d_step_new.json 2> $O/r2d_step_new.err; cut -c1-300 $O/r2d_step_new.json; tail -3 $O/r2d_step_new.err | cut -c1-300

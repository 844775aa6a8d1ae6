#!/bin/bash
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
O=gpurun_out
timeout 400 python -m pytest tests/test_conv1x1_gpu.py tests/test_fused_layer_gpu.py tests/test_head_gpu.py -m gpu -q --timeout 240 -rfE -p no:cacheprovider > $O/r2o_pytest.log 2>&1; echo "pytest rc=$?" >> $O/r2o_pytest.log
grep -E "passed|failed|^FAILED|^ERROR|rc=" $O/r2o_pytest.log | cut -c1-300 | tail -20
timeout 300 python scripts/bench_conv_abi.py --iters 20 --modes 1 --only "conv1 " > $O/r2o_conv_abi.log 2>&1; grep "conv1 " $O/r2o_conv_abi.log | cut -c1-100
timeout 300 python scripts/bench_conv_abi.py --iters 20 --modes 1 --only "conv3 " >> $O/r2o_conv_abi.log 2>&1; grep "conv3 " $O/r2o_conv_abi.log | cut -c1-100
timeout 420 python bench.py --steps 20 --warmup 8 --no-cpu-baseline --kernels new > $O/r2o_step_new.json 2> $O/r2o_step_new.err; cut -c1-200 $O/r2o_step_new.json

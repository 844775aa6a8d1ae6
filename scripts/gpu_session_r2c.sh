#!/bin/bash
# Round 2, session C: LDS 1x1 kernels with the deep (6-stage) pipeline and the LDS-transposed epilogue.
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
O=gpurun_out
timeout 300 python -m pytest tests/test_conv1x1_gpu.py tests/test_fused_layer_gpu.py tests/test_head_gpu.py tests/test_rccl_gpu.py -m gpu -q --timeout 240 -rfE -p no:cacheprovider > $O/r2c_pytest.log 2>&1; echo "pytest rc=$?" >> $O/r2c_pytest.log
grep -E "passed|failed|^FAILED|^ERROR|rc=" $O/r2c_pytest.log | cut -c1-300 | tail -20
timeout 300 python scripts/bench_conv_abi.py --iters 20 --modes 1 --json $O/r2c_conv_abi.json > $O/r2c_conv_abi.log 2>&1; tail -26 $O/r2c_conv_abi.log | cut -c1-160
timeout 100 python scripts/diag_7x7.py > $O/r2c_diag_7x7.log 2>&1; tail -18 $O/r2c_diag_7x7.log | cut -c1-220
B="timeout 420 python bench.py --steps 20 --warmup 8 --no-cpu-baseline"
$B --kernels new > $O/r2c_step_new.json 2> $O/r2c_step_new.err; cut -c1-300 $O/r2c_step_new.json; tail -3 $O/r2c_step_new.err | cut -c1-300

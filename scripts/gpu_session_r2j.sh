#!/bin/bash
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
O=gpurun_out
timeout 400 python -m pytest tests/test_conv1x1_gpu.py tests/test_conv3x3g_gpu.py tests/test_fused_layer_gpu.py tests/test_group_norm9_gpu.py tests/test_flat_sgd_gpu.py tests/test_head_gpu.py tests/test_stem_gpu.py -m gpu -q --timeout 240 -rfE -p no:cacheprovider > $O/r2j_pytest.log 2>&1; echo "pytest rc=$?" >> $O/r2j_pytest.log
grep -E "passed|failed|^FAILED|^ERROR|rc=" $O/r2j_pytest.log | cut -c1-300 | tail -20
timeout 300 python scripts/bench_conv_abi.py --iters 20 --modes 1 --only "s1 " > $O/r2j_conv_abi.log 2>&1; tail -6 $O/r2j_conv_abi.log | cut -c1-100
timeout 300 python scripts/bench_conv_abi.py --iters 20 --modes 1 --only "s2 " >> $O/r2j_conv_abi.log 2>&1; tail -6 $O/r2j_conv_abi.log | cut -c1-100
COT_TUNING=17=4 timeout 300 python scripts/bench_conv_abi.py --iters 20 --modes 1 --only "s2 " > $O/r2j_conv_abi_oldwgrad.log 2>&1; tail -6 $O/r2j_conv_abi_oldwgrad.log | cut -c1-100
B="timeout 420 python bench.py --steps 20 --warmup 8 --no-cpu-baseline"
$B --kernels new > $O/r2j_step_new.json 2> $O/r2j_step_new.err; cut -c1-200 $O/r2j_step_new.json; tail -3 $O/r2j_step_new.err | cut -c1-300

#!/bin/bash
# Round 2, session B: first hardware contact of the LDS-pipelined 1x1 kernels, the workgroup-level weight-gradient slices,
# the parallel reduce and the 2x2 max-pool backward.   gpurun --timeout 900 -- bash scripts/gpu_session_r2b.sh
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
O=gpurun_out
timeout 400 python -m pytest tests/test_conv1x1_gpu.py tests/test_pool_gpu.py tests/test_fused_layer_gpu.py tests/test_head_gpu.py tests/test_rccl_gpu.py tests/test_input_pipeline_gpu.py -m gpu -q --timeout 240 -rfE -p no:cacheprovider > $O/r2b_pytest.log 2>&1; echo "pytest rc=$?" >> $O/r2b_pytest.log
grep -E "passed|failed|^FAILED|^ERROR|rc=" $O/r2b_pytest.log | cut -c1-300 | tail -20
timeout 300 python scripts/bench_conv_abi.py --iters 20 --json $O/r2b_conv_abi.json > $O/r2b_conv_abi.log 2>&1; tail -50 $O/r2b_conv_abi.log | cut -c1-160
timeout 100 python scripts/diag_7x7.py > $O/r2b_diag_7x7.log 2>&1; tail -16 $O/r2b_diag_7x7.log | cut -c1-200
B="timeout 420 python bench.py --steps 20 --warmup 8 --no-cpu-baseline"
COT_KERNEL_SUMMARY=$O/r2b_kernels_new.json $B --kernels new > $O/r2b_step_new.json 2> $O/r2b_step_new.err; cut -c1-300 $O/r2b_step_new.json; tail -3 $O/r2b_step_new.err | cut -c1-300
mkdir -p /tmp/prof && cd /tmp/prof && COT_ROCTX=1 timeout 400 rocprofv3 --kernel-trace --marker-trace --stats --output-format csv \
    -d /tmp/prof/out -o trace -- python $GRAFT_REPO_ROOT/bench.py --kernels new --steps 5 --warmup 4 --no-cpu-baseline --no-kernel-timing \
    > $GRAFT_REPO_ROOT/$O/r2b_prof.log 2>&1
cd $GRAFT_REPO_ROOT
python scripts/trace_summary.py /tmp/prof/out --steps 5 --out $O/r2b_trace_new_per_shape.csv; head -50 $O/r2b_trace_new_per_shape.csv | cut -c1-180
du -sh $O

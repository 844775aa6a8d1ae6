#!/bin/bash
# Round 2, session A: every GPU test (no -x: each kernel family gets executed), microbenchmarks of the convolution and
# aggregation kernels, the step on both kernel sets with per-kernel summaries, and a per-shape kernel trace of `new`.
#   gpurun --timeout 1500 -- bash scripts/gpu_session_r2a.sh
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
O=gpurun_out
timeout 700 python -m pytest tests -m gpu -q --timeout 240 -rfE -p no:cacheprovider > $O/r2a_pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/r2a_pytest_gpu.log
grep -E "passed|failed|^FAILED|^ERROR" $O/r2a_pytest_gpu.log | cut -c1-250 | tail -40
timeout 200 python scripts/bench_agg_abi.py --shapes 0,1,2,3 --variants v3d --dtypes bf16,fp32 > $O/r2a_bench_agg.log 2>&1; tail -24 $O/r2a_bench_agg.log | cut -c1-200
timeout 300 python scripts/bench_conv1x1.py --iters 20 > $O/r2a_bench_conv1x1.log 2>&1; tail -45 $O/r2a_bench_conv1x1.log | cut -c1-200
timeout 200 python scripts/bench_conv3x3g.py --iters 20 > $O/r2a_bench_conv3x3g.log 2>&1; tail -12 $O/r2a_bench_conv3x3g.log | cut -c1-200
B="timeout 420 python bench.py --steps 20 --warmup 8 --no-cpu-baseline"
COT_KERNEL_SUMMARY=$O/r2a_kernels_round1.json $B --kernels round1 > $O/r2a_step_round1.json 2> $O/r2a_step_round1.err; cut -c1-400 $O/r2a_step_round1.json
COT_KERNEL_SUMMARY=$O/r2a_kernels_new.json $B --kernels new > $O/r2a_step_new.json 2> $O/r2a_step_new.err; cut -c1-400 $O/r2a_step_new.json; tail -3 $O/r2a_step_new.err | cut -c1-300
head -c 3000 $O/r2a_kernels_new.json
$B --kernels new --graph > $O/r2a_step_new_graph.json 2> $O/r2a_step_new_graph.err; cut -c1-300 $O/r2a_step_new_graph.json; tail -2 $O/r2a_step_new_graph.err | cut -c1-300
# per-shape kernel trace of the `new` set, timed region only
mkdir -p /tmp/prof && cd /tmp/prof && COT_ROCTX=1 timeout 400 rocprofv3 --kernel-trace --marker-trace --stats --output-format csv \
    -d /tmp/prof/out -o trace -- python $GRAFT_REPO_ROOT/bench.py --kernels new --steps 5 --warmup 4 --no-cpu-baseline --no-kernel-timing \
    > $GRAFT_REPO_ROOT/$O/r2a_prof.log 2>&1
cd $GRAFT_REPO_ROOT
python scripts/trace_summary.py /tmp/prof/out --steps 5 --out $O/r2a_trace_new_per_shape.csv; head -40 $O/r2a_trace_new_per_shape.csv | cut -c1-200
for f in $(find /tmp/prof/out -name "*kernel_stats*.csv"); do cp $f $O/r2a_$(basename $f); done
rm -rf /tmp/prof/out
mkdir -p /tmp/prof && cd /tmp/prof && COT_ROCTX=1 timeout 400 rocprofv3 --kernel-trace --marker-trace --stats --output-format csv \
    -d /tmp/prof/out -o trace -- python $GRAFT_REPO_ROOT/bench.py --kernels round1 --steps 5 --warmup 4 --no-cpu-baseline --no-kernel-timing \
    > $GRAFT_REPO_ROOT/$O/r2a_prof_round1.log 2>&1
cd $GRAFT_REPO_ROOT
python scripts/trace_summary.py /tmp/prof/out --steps 5 --out $O/r2a_trace_round1_per_shape.csv; head -25 $O/r2a_trace_round1_per_shape.csv | cut -c1-200
# the default invocation (auto probe with the fp32-truth gate)
timeout 900 python bench.py --steps 20 --warmup 8 > $O/r2a_step_auto.json 2> $O/r2a_step_auto.err; cut -c1-1500 $O/r2a_step_auto.json; grep "kernel set" $O/r2a_step_auto.err | cut -c1-1200
# multi-GPU launcher path on a 1-GPU box: must reach the ranks and fail only on the device count
timeout 120 python bench.py --gpus 2 --steps 2 --warmup 1 > $O/r2a_gpus2.log 2>&1; echo "gpus2 rc=$?"; tail -5 $O/r2a_gpus2.log | cut -c1-300
du -sh $O

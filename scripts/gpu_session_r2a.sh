#!/bin/bash
# Round-2 first hardware contact for everything written after round 1's GPU budget ran out (ROUND2_PLAN.md).
#   gpurun --timeout 1500 -- bash scripts/gpu_session_r2a.sh
# One call: parity of the new kernels, their microbenchmarks, then the training step in the five configurations that matter,
# and a kernel trace of the best one.  Everything lands in gpurun_out/ (small files only).
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
O=gpurun_out
# 1. parity: the whole GPU suite, new files last (pytest -x stops at the first failure)
timeout 900 python -m pytest tests -m gpu -q -x > $O/r2a_pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/r2a_pytest_gpu.log
tail -6 $O/r2a_pytest_gpu.log | cut -c1-300
# 1b. the aggregation kernels changed after their last measurement (118 -> 79 VALU per channel in the
#     bf16 backward loop); round-1 figures at N80xC64x56x56: bf16 fwd 22.0 us / bwd 44.9 us cold
timeout 200 python scripts/bench_agg_abi.py --shapes 0,1 --variants v3d --dtypes bf16,fp32 > $O/r2a_bench_agg.log 2>&1; tail -14 $O/r2a_bench_agg.log | cut -c1-200
# 2. microbenchmarks of the two convolution families against MIOpen / rocBLAS
timeout 300 python scripts/bench_conv1x1.py --iters 20 > $O/r2a_bench_conv1x1.log 2>&1; tail -45 $O/r2a_bench_conv1x1.log | cut -c1-200
timeout 200 python scripts/bench_conv3x3g.py --iters 20 > $O/r2a_bench_conv3x3g.log 2>&1; tail -12 $O/r2a_bench_conv3x3g.log | cut -c1-200
# 3. the step: default (MIOpen convolutions) / hip 1x1 / hip 1x1 + 3x3 / + single-node CotLayer / + HIP graph
B="timeout 420 python bench.py --steps 20 --warmup 8 --no-cpu-baseline --kernels round1"
$B                                              > $O/r2a_step_default.json   2> $O/r2a_step_default.err;   cut -c1-700 $O/r2a_step_default.json
$B --conv1x1 hip                                > $O/r2a_step_c1.json        2> $O/r2a_step_c1.err;        cut -c1-700 $O/r2a_step_c1.json
$B --conv1x1 hip --conv3x3 hip --gn9            > $O/r2a_step_c1c3.json      2> $O/r2a_step_c1c3.err;      cut -c1-700 $O/r2a_step_c1c3.json
COT_KERNEL_SUMMARY=$O/r2a_kernels_fused.json $B --fused-layer > $O/r2a_step_fused.json 2> $O/r2a_step_fused.err; cut -c1-700 $O/r2a_step_fused.json
head -c 2500 $O/r2a_kernels_fused.json
COT_TUNING=12=1 $B --fused-layer                > $O/r2a_step_fused_bnfold.json 2> $O/r2a_step_fused_bnfold.err; cut -c1-700 $O/r2a_step_fused_bnfold.json
$B --fused-layer --graph                        > $O/r2a_step_fused_graph.json 2> $O/r2a_step_fused_graph.err; cut -c1-700 $O/r2a_step_fused_graph.json
tail -3 $O/r2a_step_fused.err | cut -c1-300
timeout 900 python bench.py --steps 20 --warmup 8 --no-cpu-baseline > $O/r2a_step_auto.json 2> $O/r2a_step_auto.err; cut -c1-900 $O/r2a_step_auto.json; grep "kernel set" $O/r2a_step_auto.err | cut -c1-700
# 4. which op breaks graph replay
timeout 200 python scripts/graph_bisect.py > $O/r2a_graph_bisect.log 2>&1; cut -c1-200 $O/r2a_graph_bisect.log | tail -22
# 5. kernel trace of the fused configuration, timed region only
mkdir -p /tmp/prof && cd /tmp/prof && COT_ROCTX=1 timeout 400 rocprofv3 --kernel-trace --marker-trace --stats --output-format csv \
    -d /tmp/prof/out -o trace -- python $GRAFT_REPO_ROOT/bench.py --fused-layer --steps 5 --warmup 3 --no-cpu-baseline --no-kernel-timing \
    > $GRAFT_REPO_ROOT/$O/r2a_prof.log 2>&1
cd $GRAFT_REPO_ROOT
for f in $(find /tmp/prof/out -name "*kernel_stats*.csv" -o -name "*marker*stats*.csv"); do cp $f $O/r2a_$(basename $f); done
head -30 $O/r2a_trace_kernel_stats.csv | cut -c1-170
du -sh $O

#!/bin/bash
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 330 python bench.py --steps 20 --warmup 5 > gpurun_out/bench_final.json 2> gpurun_out/bench_final.err; cut -c1-1800 gpurun_out/bench_final.json; tail -2 gpurun_out/bench_final.err
mkdir -p /tmp/prof && cd /tmp/prof && COT_ROCTX=1 timeout 300 rocprofv3 --kernel-trace --marker-trace --stats --output-format csv -d /tmp/prof/out -o trace -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 3 --no-cpu-baseline --no-kernel-timing > $GRAFT_REPO_ROOT/gpurun_out/prof_train.log 2>&1
cd $GRAFT_REPO_ROOT
for f in $(find /tmp/prof/out -name "*stats*.csv"); do cp $f gpurun_out/; done
head -8 gpurun_out/trace_kernel_stats.csv | cut -c1-150
DET=1 B=8 timeout 200 python scripts/debug_graph.py 2>&1 | grep -E "deterministic|eager loss|replay" | cut -c1-300

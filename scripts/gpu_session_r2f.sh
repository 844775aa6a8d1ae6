#!/bin/bash
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
O=gpurun_out
python scripts/ubench_trprobe.py > $O/r2_trprobe.log 2>&1; head -40 $O/r2_trprobe.log
mkdir -p /tmp/prof && cd /tmp/prof && COT_ROCTX=1 timeout 400 rocprofv3 --kernel-trace --marker-trace --stats --output-format csv \
    -d /tmp/prof/out -o trace -- python $GRAFT_REPO_ROOT/bench.py --kernels new --steps 5 --warmup 4 --no-cpu-baseline --no-kernel-timing \
    > $GRAFT_REPO_ROOT/$O/r2f_prof.log 2>&1
cd $GRAFT_REPO_ROOT
python scripts/trace_summary.py /tmp/prof/out --steps 5 --out $O/r2f_trace_new_per_shape.csv; head -5 $O/r2f_trace_new_per_shape.csv | cut -c1-180

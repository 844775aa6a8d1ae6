#!/bin/bash
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -q -x > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
tail -4 gpurun_out/pytest_gpu.log | cut -c1-300
timeout 400 python bench.py --steps 20 --warmup 5 > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; cut -c1-2500 gpurun_out/bench_default.json
mkdir -p /tmp/prof && cd /tmp/prof && COT_ROCTX=1 timeout 500 rocprofv3 --kernel-trace --marker-trace --stats --output-format csv -d /tmp/prof/out -o trace -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 3 --no-cpu-baseline --no-kernel-timing > $GRAFT_REPO_ROOT/gpurun_out/prof_train.log 2>&1
cd $GRAFT_REPO_ROOT
tail -3 gpurun_out/prof_train.log | cut -c1-300
for f in $(find /tmp/prof/out -name "*stats*.csv"); do cp $f gpurun_out/; done
head -25 gpurun_out/trace_kernel_stats.csv | cut -c1-160
ls -la gpurun_out; du -sh gpurun_out

#!/bin/bash
# session 2: finish GPU tests, v1/v2 kernel A/B, rocprof kernel stats of one training step (small outputs only)
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
tail -25 gpurun_out/pytest_gpu.log
timeout 500 python scripts/bench_agg_abi.py --iters 20 --rounds 3 --variants v1,v2dpp,v2shfl,v2dpp_fP4,v2dpp_bP2 --out gpurun_out/agg_ab.json > gpurun_out/agg_ab.log 2>&1; cat gpurun_out/agg_ab.log
mkdir -p /tmp/prof && cd /tmp/prof && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof/out -o trace -- python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 2 --dtype bf16 --layout nchw --no-cpu-baseline --no-kernel-timing > $GRAFT_REPO_ROOT/gpurun_out/prof_train.log 2>&1
cd $GRAFT_REPO_ROOT
find /tmp/prof/out -type f | head -20
for f in $(find /tmp/prof/out -name "*stats*.csv"); do cp $f gpurun_out/$(basename $f); done
ls -la gpurun_out
f=$(ls gpurun_out/*kernel_stats.csv | head -1); [ -n "$f" ] && head -60 "$f" | cut -c1-220
du -sh gpurun_out

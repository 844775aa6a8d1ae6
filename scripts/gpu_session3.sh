#!/bin/bash
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -q -x > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
tail -6 gpurun_out/pytest_gpu.log | cut -c1-300
timeout 400 python bench.py --steps 10 --warmup 3 > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; cut -c1-1500 gpurun_out/bench_default.json
timeout 300 python scripts/profile_step.py --out gpurun_out/torch_prof.txt > /dev/null 2> gpurun_out/torch_prof.err; cat gpurun_out/torch_prof.txt | cut -c1-200
mkdir -p /tmp/prof && cd /tmp/prof && timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof/out -o trace -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-kernel-timing > $GRAFT_REPO_ROOT/gpurun_out/prof_train.log 2>&1
cd $GRAFT_REPO_ROOT
find /tmp/prof/out -type f | head; for f in $(find /tmp/prof/out -name "*stats*.csv"); do cp $f gpurun_out/; done
ls -la gpurun_out; du -sh gpurun_out

#!/bin/bash
# where do the ~200 __amd_rocclr_copyBuffer launches per replayed step come from?
mkdir -p gpurun_out; export TMPDIR=/tmp
rm -rf /tmp/prof/out; mkdir -p /tmp/prof && cd /tmp/prof && COT_ROCTX=1 timeout 400 rocprofv3 --kernel-trace --marker-trace --output-format csv \
    -d /tmp/prof/out -o trace -- python $GRAFT_REPO_ROOT/bench.py --kernels new --steps 5 --warmup 4 --no-cpu-baseline --no-kernel-timing --no-secondary --no-pmc "$@" > /tmp/prof/log 2>&1
cd $GRAFT_REPO_ROOT
python scripts/trace_neighbours.py /tmp/prof/out copyBuffer --top 40 --all | tee gpurun_out/r06_copybuffer.log
head -2 $(find /tmp/prof/out -name "*kernel_trace.csv" | head -1)

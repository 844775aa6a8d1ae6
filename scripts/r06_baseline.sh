#!/bin/bash
# round-6 baseline session: default bench line + single-stream eager kernel trace of the step (per-shape CSV)
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
O=gpurun_out
T0=$(date +%s)
timeout 900 python bench.py > $O/r06a_bench_default.json 2> $O/r06a_bench_default.err; echo "bench rc=$? wall=$(( $(date +%s) - T0 ))s"
cut -c1-300 $O/r06a_bench_default.json
COT_WGRAD_STREAM=0 bash scripts/gpu_trace_new.sh r06a_single --no-pmc --eager > $O/r06a_trace_sh.log 2>&1; tail -2 $O/r06a_trace_sh.log | cut -c1-200
echo "session wall=$(( $(date +%s) - T0 ))s"

#!/bin/bash
mkdir -p gpurun_out
O=gpurun_out
T0=$(date +%s)
timeout 1500 python bench.py > $O/r2n_bench_default.json 2> $O/r2n_bench_default.err
echo "bench rc=$? wall=$(( $(date +%s) - T0 ))s"
cut -c1-3000 $O/r2n_bench_default.json; grep -E "kernel set" $O/r2n_bench_default.err | cut -c1-1800

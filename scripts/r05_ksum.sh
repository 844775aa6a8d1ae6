#!/bin/bash
# usage: r05_ksum.sh TAG "name:ENV=VAL[,ENV=VAL]:bench flags" ...  -- per-kernel device time of the instrumented steps (COT_KERNEL_SUMMARY)
mkdir -p gpurun_out; export TMPDIR=/tmp; O=gpurun_out
TAG=$1; shift
for t in "$@"; do
  name=${t%%:*}; rest=${t#*:}; envs=${rest%%:*}; flags=${rest#*:}
  ( for kv in ${envs//,/ }; do export $kv; done
    COT_KERNEL_SUMMARY=$O/${TAG}_ksum_$name.json timeout 300 python bench.py --kernels new $flags --steps 5 --warmup 3 --settle-seconds 3 --no-cpu-baseline --no-secondary --no-pmc > $O/${TAG}_ksum_${name}_line.json 2> $O/${TAG}_ksum_$name.err || tail -3 $O/${TAG}_ksum_$name.err )
done

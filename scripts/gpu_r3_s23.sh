#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp; O=gpurun_out
for t in "25=64" "25=64,24=1" "25=0,24=1" "25=64,24=4" "25=64,24=5"; do
  echo "== tune $t" >> $O/r3s23_wgrad_ab.log
  timeout 300 python scripts/bench_conv_abi.py --modes 1 --tune "$t" --only conv1 2>&1 | grep "^s[0-9e]" | awk -F'|' '{print substr($1,1,30) "|" $3}' >> $O/r3s23_wgrad_ab.log
done

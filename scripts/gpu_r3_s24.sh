#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp; O=gpurun_out
for t in "25=0" "25=64" "25=64,24=1"; do
for b in 40 80 160 320; do
  echo "== tune $t batch $b" >> $O/r3s24_wgrad_n.log
  timeout 300 python scripts/bench_conv_abi.py --modes 1 --tune "$t" --batch $b --only "conv1 " 2>&1 | grep "^s[0-9e]" | awk -F'|' '{print substr($1,1,30) "|" $3}' >> $O/r3s24_wgrad_n.log
done; done

#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp; O=gpurun_out
timeout 200 python scripts/check_wgrad_variant.py 8 2>&1 | tail -1 >> $O/r3s43.log
for t in "25=0" "25=8" "25=0" "25=8"; do
  echo "== tune $t" >> $O/r3s43.log
  timeout 300 python scripts/bench_conv_abi.py --modes 1 --tune "$t" --only "conv" 2>&1 | grep "^s[0-9]" | awk -F'|' '{print substr($1,1,30) "|" $3}' >> $O/r3s43.log
done

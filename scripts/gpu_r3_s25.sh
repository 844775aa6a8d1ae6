#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp; O=gpurun_out
for v in 128 192 64; do timeout 200 python scripts/check_wgrad_variant.py $v 2>&1 | tail -4 >> $O/r3s25_check.log; done
for t in "25=0" "25=128" "25=192" "25=0" "25=128"; do
  echo "== tune $t" >> $O/r3s25_wgrad_ab.log
  timeout 300 python scripts/bench_conv_abi.py --modes 1 --tune "$t" 2>&1 | grep "^s[0-9e]" | awk -F'|' '{print substr($1,1,30) "|" $3}' >> $O/r3s25_wgrad_ab.log
done

#!/bin/bash
# round 3, session 6: weight gradient with the wide reduce: slice-count policy sweep (key 25: bits 8.. cap %, bits 16.. workgroups per CU x 4)
mkdir -p gpurun_out; export TMPDIR=/tmp; O=gpurun_out
for t in "25=1" "25=16" "25=0" "25=262144" "25=25600" "25=287744" "25=51200" "25=313344" "25=524288" "25=575488"; do
  echo "== tune $t" >> $O/r3s6_wgrad_ab.log
  timeout 300 python scripts/bench_conv_abi.py --modes 1 --tune "$t" 2>&1 | grep "^s[0-9e]" | awk -F'|' '{print substr($1,1,30) "|" $3}' >> $O/r3s6_wgrad_ab.log
done

#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp; O=gpurun_out
for t in "25=0" "25=64" "25=128" "25=192" "25=0" "25=64"; do
  echo "== tune $t" >> $O/r3s22_wgrad_ab.log
  timeout 300 python scripts/bench_conv_abi.py --modes 1 --tune "$t" 2>&1 | grep "^s[0-9e]" | awk -F'|' '{print substr($1,1,30) "|" $3}' >> $O/r3s22_wgrad_ab.log
done
timeout 200 python -m pytest tests/test_conv1x1_gpu.py -m gpu -x -q 2>&1 | tail -2 >> $O/r3s22_wgrad_ab.log

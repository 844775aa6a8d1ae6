#!/bin/bash
# round 3, session 5: timing ablations of the third-generation weight gradient + stage depth
mkdir -p gpurun_out; export TMPDIR=/tmp; O=gpurun_out
for cfg in "24=0" "24=1" "24=2" "24=4" "24=8" "24=16" "24=7" "24=31" "25=8" "25=25608" "25=262152"; do
  echo "== $cfg" >> $O/r3s5_wg_ablate.log
  for sh in "s4 conv1 " "s3 conv1 " "s1 conv1 " "s2 conv3 "; do
    timeout 120 python scripts/bench_conv_abi.py --modes 1 --only "$sh" --tune "$cfg" 2>&1 | grep "^s[0-9]" | awk -F'|' '{print $1 "|" $3}' >> $O/r3s5_wg_ablate.log
  done
done
cat $O/r3s5_wg_ablate.log

#!/bin/bash
# round 3, session 3: timing ablations of the third-generation 1x1 kernel (cot_set_tuning key 24) on three layer shapes
mkdir -p gpurun_out; export TMPDIR=/tmp; O=gpurun_out
for ab in 0 1 2 4 8 16 6 7 15 31; do
  echo "== ablate $ab" >> $O/r3s3_ablate.log
  for sh in "s4 conv1 " "s3 conv1 " "s1 conv1 " "s2 conv1 "; do
    timeout 120 python scripts/bench_conv_abi.py --modes 1 --only "$sh" --tune "24=$ab" 2>&1 | grep "^s[0-9]" >> $O/r3s3_ablate.log
  done
done
cat $O/r3s3_ablate.log

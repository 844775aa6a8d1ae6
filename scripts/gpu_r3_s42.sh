#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp; O=gpurun_out
for v in 8 0; do timeout 200 python scripts/check_fwd_variant.py $v 2>&1 | tail -2 >> $O/r3s42_check.log; done
for t in "23=0" "23=8" "23=0" "23=8"; do
  echo "== tune $t" >> $O/r3s42_fwd_ab.log
  timeout 300 python scripts/bench_conv_abi.py --modes 1 --tune "$t" 2>&1 | grep "^s[0-9]" | cut -c1-86 >> $O/r3s42_fwd_ab.log
done

#!/bin/bash
# round 3, session 1: hardware probes + where the copyBuffer launches come from + images-per-workgroup sweep on the 7x7 layers
mkdir -p gpurun_out; export TMPDIR=/tmp; O=gpurun_out
timeout 120 python scripts/probe3.py > $O/r3s1_probe3.log 2>&1; echo "probe rc=$?" >> $O/r3s1_probe3.log
timeout 400 python scripts/diag_copybuffer.py > $O/r3s1_copybuffer.log 2>&1; echo "diag rc=$?" >> $O/r3s1_copybuffer.log
for t in "" "16=1" "16=2" "16=3" "16=4"; do
  echo "== tune $t" >> $O/r3s1_ni_sweep.log
  timeout 200 python scripts/bench_conv_abi.py --only "s4" --modes 1 --tune "$t" >> $O/r3s1_ni_sweep.log 2>&1
done
tail -5 $O/r3s1_probe3.log; tail -3 $O/r3s1_copybuffer.log

#!/bin/bash
# per-dispatch durations of the weight-gradient pair (main kernel + reduce) per shape
mkdir -p gpurun_out; cd /tmp; export TMPDIR=/tmp; R=/root/repo; O=$R/gpurun_out
for sh in "s1 conv1" "s2 conv1 " "s3 conv1 " "s3 conv1x1" "s4 conv1 " "s4 conv3" "s4 conv1x1"; do
  rm -rf /tmp/kt; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/kt -o kt --output-format csv -- python $R/scripts/bench_conv_abi.py --modes 1 --only "$sh" --iters 20 > /tmp/kt.log 2>&1
  f=$(find /tmp/kt -name "*kernel_stats.csv" | head -1)
  echo "== $sh" >> $O/r3s20_wgrad_pair.txt
  grep "^s[0-9]" /tmp/kt.log | cut -c1-110 >> $O/r3s20_wgrad_pair.txt
  python - "$f" >> $O/r3s20_wgrad_pair.txt <<'P'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    n=r['Name']
    if 'wgrad' in n or 'reduce' in n: print(f"   {n[:70]:70s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:8.2f} us  min {float(r['MinNs'])/1e3:8.2f}")
P
done

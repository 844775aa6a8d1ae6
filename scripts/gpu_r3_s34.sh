#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out
cd /tmp
for n in 4 14; do
rm -rf /tmp/ht$n; timeout 300 rocprofv3 --hip-runtime-trace --stats --output-format csv -d /tmp/ht$n -o t -- python $GRAFT_REPO_ROOT/scripts/step_plain.py $n > /tmp/ht$n.log 2>&1
f=$(find /tmp/ht$n -name "*hip_api_stats.csv" | head -1)
echo "== steps $n" >> $O/r3s34_api.txt; grep -i "memcpy\|LaunchKernel\|EventRecord\|Malloc" $f | cut -d, -f1,2,3,4 >> $O/r3s34_api.txt
done

#!/bin/bash
# which HIP API calls does a training step make (and which of them become __amd_rocclr_copyBuffer launches)?
mkdir -p gpurun_out; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out
rm -rf /tmp/ht; mkdir -p /tmp/ht; cd /tmp/ht
COT_ROCTX=1 timeout 400 rocprofv3 --hip-runtime-trace --kernel-trace --marker-trace --stats --output-format csv -d /tmp/ht/out -o t -- python $GRAFT_REPO_ROOT/bench.py --kernels new --steps 5 --warmup 4 --no-cpu-baseline --no-kernel-timing --settle-seconds 0 > $O/r3s33_prof.log 2>&1
for f in $(find /tmp/ht/out -name "*stats*.csv"); do echo "== $f"; head -25 $f | cut -c1-160; done > $O/r3s33_stats.txt
f=$(find /tmp/ht/out -name "*hip_api_trace.csv" | head -1)
python - "$f" >> $O/r3s33_stats.txt <<'P'
import csv,sys,collections
c=collections.Counter()
for r in csv.DictReader(open(sys.argv[1])): c[r.get('Function','?')]+=1
print("== HIP API calls in the profiled window (5 steps)")
for k,v in c.most_common(20): print(f"{v:8d} {v/5:9.1f}/step  {k}")
P

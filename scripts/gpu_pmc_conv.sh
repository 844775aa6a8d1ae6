#!/bin/bash
# SQ counters of the LDS 1x1 kernel on one deep-K shape: where do the wave cycles go?
set -x
mkdir -p gpurun_out /tmp/pmc
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
cd /tmp/pmc
rocprofv3 -L 2>/dev/null | grep -E "SQ_(WAVE_CYCLES|WAIT|ACTIVE_INST|INSTS_|LDS|BUSY)" | cut -c1-120 | head -60 > $O/pmc_conv_list.txt
for SHAPE in "s4 conv1" "s3 conv1" "s1 conv3"; do
  TAG=$(echo $SHAPE | tr ' ' '_')
  timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS \
      --output-format csv -d /tmp/pmc/out_$TAG -o p -- python $GRAFT_REPO_ROOT/scripts/bench_conv_abi.py --iters 3 --modes 1 --only "$SHAPE" > $O/pmc_conv_$TAG.log 2>&1
  f=$(find /tmp/pmc/out_$TAG -name "*counter_collection.csv" | head -1)
  python - "$f" "$O/pmc_conv_$TAG.csv" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in rows:
    k = r["Kernel_Name"][:60]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); 
    if r["Counter_Name"] == "SQ_WAVE_CYCLES": cnt[k] += 1
with open(sys.argv[2], "w") as f:
    for k, d in agg.items():
        f.write(k + "," + str(cnt[k]) + "," + ",".join(f"{n}={v/max(cnt[k],1):.0f}" for n, v in sorted(d.items())) + "\n")
PY
  cat $O/pmc_conv_$TAG.csv | grep -i "lds_fwd\|wgrad" | cut -c1-400
done

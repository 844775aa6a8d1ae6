#!/bin/bash
# SQ counters of the third-generation 1x1 kernels (forward / data gradient / weight gradient) on two deep shapes, three passes
mkdir -p gpurun_out /tmp/pmc
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
cd /tmp/pmc
P1="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS"
P2="SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_ACTIVE_INST_MISC SQ_INSTS_VALU SQ_INSTS_SALU"
P3="SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_FLAT SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_WAVES"
for SHAPE in "s3 conv1 " "s4 conv1 " "s2 conv1 "; do
  TAG=$(echo $SHAPE | tr -d ' ')
  i=0
  for P in "$P1" "$P2" "$P3"; do
    i=$((i+1))
    timeout 200 rocprofv3 --kernel-trace --pmc $P --output-format csv -d /tmp/pmc/out_${TAG}_$i -o p -- python $GRAFT_REPO_ROOT/scripts/bench_conv_abi.py --iters 3 --modes 1 --only "$SHAPE" > $O/pmc3_$TAG_$i.log 2>&1
    f=$(find /tmp/pmc/out_${TAG}_$i -name "*counter_collection.csv" | head -1)
    python - "$f" "$TAG pass$i" >> $O/r3_pmc_conv3.txt <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(collections.Counter)
for r in rows:
    k = r["Kernel_Name"][:70]
    if "lds_fwd2" not in k and "wgrad_lds2" not in k: continue
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k][r["Counter_Name"]] += 1
for k, d in agg.items():
    print(sys.argv[2], "|", k, "|", " ".join(f"{n}={v/max(cnt[k][n],1):.0f}" for n, v in sorted(d.items())))
PY
  done
done
cat $O/r3_pmc_conv3.txt | cut -c1-600

#!/bin/bash
# memory-side counters of the third-generation 1x1 kernels: L2 hits / misses, fabric-side fetch and write bytes
mkdir -p gpurun_out /tmp/pmc
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
cd /tmp/pmc
P1="TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum"
P2="FETCH_SIZE WRITE_SIZE"
P3="TCP_TCC_READ_REQ_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum"
for SHAPE in "s3 conv1 " "s4 conv1 " "s2 conv1 " "s1 conv1 "; do
  TAG=$(echo $SHAPE | tr -d ' ')
  i=0
  for P in "$P1" "$P2" "$P3"; do
    i=$((i+1))
    timeout 200 rocprofv3 --kernel-trace --pmc $P --output-format csv -d /tmp/pmc/m_${TAG}_$i -o p -- python $GRAFT_REPO_ROOT/scripts/bench_conv_abi.py --iters 3 --modes 1 --only "$SHAPE" > $O/pmcm_${TAG}_$i.log 2>&1
    f=$(find /tmp/pmc/m_${TAG}_$i -name "*counter_collection.csv" | head -1)
    python - "$f" "$TAG pass$i" >> $O/r3_pmc_conv_mem.txt <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(collections.Counter)
for r in rows:
    k = r["Kernel_Name"][:70]
    if "lds_fwd2" not in k and "wgrad" not in k: continue
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k][r["Counter_Name"]] += 1
for k, d in agg.items():
    print(sys.argv[2], "|", k, "|", " ".join(f"{n}={v/max(cnt[k][n],1):.0f}" for n, v in sorted(d.items())))
PY
  done
done

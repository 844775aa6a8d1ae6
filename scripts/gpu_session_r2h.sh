#!/bin/bash
# aggregation kernels: knob A/B (XCD order, split backward, phase size, P, waves) + SQ counters of the default
set -x
mkdir -p gpurun_out /tmp/pmc
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
timeout 200 python scripts/bench_agg_abi.py --shapes 0,1 --dtypes bf16 --rounds 3 --variants v3d,v3d_xcd,v3d_split,v3d_xcd_split,v3d_jp8,v3d_jp2,v3d_bP4,v3d_nw8,v3d_pad32 > $O/r2h_agg_ab.log 2>&1; grep -v "max|diff" $O/r2h_agg_ab.log | cut -c1-190
cd /tmp/pmc
timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS \
      --output-format csv -d /tmp/pmc/out_agg -o p -- python $GRAFT_REPO_ROOT/scripts/bench_agg_abi.py --shapes 0 --dtypes bf16 --rounds 1 --iters 4 --variants v3d > $O/r2h_pmc_agg.log 2>&1
f=$(find /tmp/pmc/out_agg -name "*counter_collection.csv" | head -1)
python - "$f" "$O/r2h_pmc_agg.csv" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in rows:
    k = r["Kernel_Name"][:70]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "SQ_WAVE_CYCLES": cnt[k] += 1
with open(sys.argv[2], "w") as f:
    for k, d in agg.items():
        f.write(k + "," + str(cnt[k]) + "," + ",".join(f"{n}={v/max(cnt[k],1):.0f}" for n, v in sorted(d.items())) + "\n")
PY
grep -i "agg_" $O/r2h_pmc_agg.csv | cut -c1-400
cd $GRAFT_REPO_ROOT
for NI in 1 2 3; do echo "ni=$NI"; COT_TUNING=16=$NI timeout 100 python scripts/bench_conv_abi.py --iters 20 --modes 1 --only "s4 " 2>&1 | grep "^s4" | cut -c1-100; done > $O/r2h_ni.log 2>&1; cat $O/r2h_ni.log

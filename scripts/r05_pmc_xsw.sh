#!/bin/bash
# SQ / LDS counters of the BIG-tile 1x1 kernel on the channel-major deep layers, X stage permuted (48=1) or not (48=0)
mkdir -p gpurun_out /tmp/pmc; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out
cd /tmp/pmc
for v in 0 1; do
  timeout 250 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS \
      --output-format csv -d /tmp/pmc/out_$v -o p -- python $GRAFT_REPO_ROOT/scripts/probe_cnhw.py 3 48=$v > $O/pmc_xsw_$v.log 2>&1
  f=$(find /tmp/pmc/out_$v -name "*counter_collection.csv" | head -1)
  python - "$f" "$O/pmc_xsw_$v.csv" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in rows:
    if "lds_fwd2" not in r["Kernel_Name"]: continue
    k = r["Kernel_Name"][:48] + " grid " + r.get("Grid_Size", "?")
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "SQ_WAVE_CYCLES": cnt[k] += 1
with open(sys.argv[2], "w") as f:
    for k, d in sorted(agg.items()):
        f.write(k + "," + str(cnt[k]) + "," + ",".join(f"{n}={v/max(cnt[k],1):.0f}" for n, v in sorted(d.items())) + "\n")
PY
  echo "== 48=$v"; cut -c1-420 $O/pmc_xsw_$v.csv
done

#!/bin/bash
# LDS counters of every kernel of the training step (eager, 3 steps): which kernels spend LDS cycles on bank conflicts
mkdir -p gpurun_out /tmp/pmc; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out
cd /tmp/pmc
timeout 400 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_LDS_ADDR_CONFLICT \
    --output-format csv -d /tmp/pmc/out_step -o p -- python $GRAFT_REPO_ROOT/bench.py --kernels new --eager --steps 3 --warmup 2 --settle-seconds 0 --no-cpu-baseline --no-kernel-timing --no-secondary --no-pmc > $O/pmc_step.log 2>&1
f=$(find /tmp/pmc/out_step -name "*counter_collection.csv" | head -1)
python - "$f" "$O/pmc_step_lds.csv" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in rows:
    k = r["Kernel_Name"][:70] + " grid " + r.get("Grid_Size", "?")
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "SQ_WAVE_CYCLES": cnt[k] += 1
out = sorted(agg.items(), key=lambda kv: -kv[1].get("SQ_LDS_BANK_CONFLICT", 0))
with open(sys.argv[2], "w") as f:
    f.write("kernel,launches,conflict_total,idx_active_total,conflict_frac,busy_total,per_launch...\n")
    for k, d in out:
        c, ia = d.get("SQ_LDS_BANK_CONFLICT", 0), d.get("SQ_LDS_IDX_ACTIVE", 0)
        f.write(f"{k},{cnt[k]},{c:.0f},{ia:.0f},{(c / ia if ia else 0):.2f},{d.get('SQ_BUSY_CYCLES', 0):.0f}," + ",".join(f"{n}={v/max(cnt[k],1):.0f}" for n, v in sorted(d.items())) + "\n")
PY
head -45 $O/pmc_step_lds.csv | cut -c1-260

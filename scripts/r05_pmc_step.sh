#!/bin/bash
# counters of every kernel of the training step (eager, 3 steps).  $1 = tag, rest = the counters of the pass
mkdir -p gpurun_out /tmp/pmc; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out
TAG=$1; shift
cd /tmp/pmc
timeout 400 rocprofv3 --kernel-trace --pmc "$@" \
    --output-format csv -d /tmp/pmc/out_$TAG -o p -- python $GRAFT_REPO_ROOT/bench.py --kernels new --eager --steps 3 --warmup 2 --settle-seconds 0 --no-cpu-baseline --no-kernel-timing --no-secondary --no-pmc > $O/pmc_step_$TAG.log 2>&1
f=$(find /tmp/pmc/out_$TAG -name "*counter_collection.csv" | head -1)
python - "$f" "$O/pmc_step_$TAG.csv" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(collections.Counter)
for r in rows:
    k = r["Kernel_Name"][:90] + " grid " + r.get("Grid_Size", "?")
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k][r["Counter_Name"]] += 1
with open(sys.argv[2], "w") as f:
    for k, d in sorted(agg.items(), key=lambda kv: -sum(kv[1].values())):
        n = max(cnt[k].values())
        f.write(f"{k},{n}," + ",".join(f"{c}={v/n:.0f}" for c, v in sorted(d.items())) + "\n")
PY
head -70 $O/pmc_step_$TAG.csv | cut -c1-230

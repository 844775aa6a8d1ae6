#!/bin/bash
# first GPU session: correctness, kernel microbench, model throughput in 4 modes, rocprof kernel trace
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "import torch; print(torch.__version__, torch.cuda.get_device_name(0)); import os; print('cores', os.cpu_count())" > gpurun_out/env.txt 2>&1
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
tail -15 gpurun_out/pytest_gpu.log
timeout 300 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1; tail -3 gpurun_out/smoke.log
timeout 600 python scripts/bench_agg.py --iters 30 --out gpurun_out/bench_agg.json > gpurun_out/bench_agg.log 2>&1; cat gpurun_out/bench_agg.log
for cfg in "bf16 nchw" "bf16 nhwc" "fp32 nchw" "fp32 nhwc"; do
  set -- $cfg
  timeout 600 python bench.py --steps 10 --warmup 3 --dtype $1 --layout $2 --no-cpu-baseline > gpurun_out/bench_$1_$2.json 2> gpurun_out/bench_$1_$2.err
  python - <<PY
import json
try:
    d=json.load(open("gpurun_out/bench_$1_$2.json")); r=d.get("roofline") or {}
    print("$1 $2", d["value"], "img/s", d["ms_per_step"], "ms/step agg_share", r.get("agg_share_of_step"))
    for k in (r.get("kernels") or [])[:8]: print("   ", k)
except Exception as e:
    print("$1 $2 FAILED", e); print(open("gpurun_out/bench_$1_$2.err").read()[-2000:])
PY
done
timeout 600 python bench.py --steps 10 --warmup 3 --dtype bf16 --layout nchw --mode fwd --no-cpu-baseline > gpurun_out/bench_fwd_bf16_nchw.json 2>/dev/null; cat gpurun_out/bench_fwd_bf16_nchw.json | cut -c1-400
cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_bf16_nchw -o trace -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --dtype bf16 --layout nchw --no-cpu-baseline --no-kernel-timing > $GRAFT_REPO_ROOT/gpurun_out/prof_bf16_nchw.log 2>&1
cd $GRAFT_REPO_ROOT
find gpurun_out/prof_bf16_nchw -name "*stats*" | head; 
f=$(find gpurun_out/prof_bf16_nchw -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -40 "$f"
# keep only the small summaries
find gpurun_out/prof_bf16_nchw -name "*trace.csv" -size +20M -delete
du -sh gpurun_out

#!/bin/bash
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_flat_sgd_gpu.py tests/test_agg_gpu.py -m gpu -q -x > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
tail -12 gpurun_out/pytest_gpu.log | cut -c1-300
timeout 300 python scripts/bench_agg_abi.py --iters 20 --rounds 3 --shapes 0,1 --variants v3d,v3d_nw8,v3d_pad16,v3d_pad32,v3d_pad56,v3d_nw8_pad32,v3d_bP4,v3d_bP4_pad32 --out gpurun_out/agg_ab_knobs.json > gpurun_out/agg_ab_knobs.log 2>&1; grep -v "max|diff" gpurun_out/agg_ab_knobs.log | cut -c1-250
for cfg in "mixed nchw" "autocast nchw" "mixed nhwc"; do
  set -- $cfg
  timeout 400 python bench.py --steps 15 --warmup 4 --precision $1 --layout $2 --no-cpu-baseline > gpurun_out/bench_$1_$2.json 2> gpurun_out/bench_$1_$2.err
  python - <<PY
import json
try:
    d=json.load(open("gpurun_out/bench_$1_$2.json")); r=d.get("roofline") or {}
    print("$1 $2", d["value"], "img/s", d["ms_per_step"], "ms/step loss", d["final_loss"], "agg_share", r.get("agg_share_of_step"))
except Exception as e:
    print("$1 $2 FAILED", e); print(open("gpurun_out/bench_$1_$2.err").read()[-1500:])
PY
done
timeout 300 python scripts/profile_step.py --out gpurun_out/torch_prof_mixed.txt --mixed > /dev/null 2> gpurun_out/torch_prof.err; head -45 gpurun_out/torch_prof_mixed.txt | cut -c1-180
du -sh gpurun_out

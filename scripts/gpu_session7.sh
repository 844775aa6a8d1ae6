#!/bin/bash
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
tail -12 gpurun_out/pytest_gpu.log | cut -c1-300
run() { # name, env, args
  env $2 timeout 400 python bench.py --steps 15 --warmup 4 --no-cpu-baseline $3 > gpurun_out/bench_$1.json 2> gpurun_out/bench_$1.err
  python - <<PY
import json
try:
    d=json.load(open("gpurun_out/bench_$1.json")); r=d.get("roofline") or {}
    print("$1", d["value"], "img/s", d["ms_per_step"], "ms/step loss", d["final_loss"], "agg_share", r.get("agg_share_of_step"))
except Exception as e:
    print("$1 FAILED", e); print(open("gpurun_out/bench_$1.err").read()[-1200:])
PY
}
run mixed_fusedbn "COT_FUSED_BN=1" "--precision mixed"
run mixed_torchbn "COT_FUSED_BN=0" "--precision mixed"
run autocast_fusedbn "COT_FUSED_BN=1" "--precision autocast"
run mixed_fusedbn_nhwc "COT_FUSED_BN=1" "--precision mixed --layout nhwc"
run mixed_fusedbn_b128 "COT_FUSED_BN=1" "--precision mixed --batch 128"
timeout 300 python scripts/profile_step.py --out gpurun_out/torch_prof_mixed.txt --mixed > /dev/null 2> gpurun_out/torch_prof.err; head -60 gpurun_out/torch_prof_mixed.txt | cut -c1-180
du -sh gpurun_out

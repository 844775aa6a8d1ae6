#!/bin/bash
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 400 python -m pytest tests/test_layers_gpu.py tests/test_fused_bn_gpu.py tests/test_flat_sgd_gpu.py -m gpu -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
tail -5 gpurun_out/pytest_gpu.log | cut -c1-300
run() { # name, env, args
  env $2 timeout 400 python bench.py --steps 20 --warmup 5 $3 > gpurun_out/bench_$1.json 2> gpurun_out/bench_$1.err
  python - <<PY
import json
try:
    d=json.load(open("gpurun_out/bench_$1.json")); r=d.get("roofline") or {}
    print("$1", d["value"], "img/s", d["ms_per_step"], "ms/step loss", d["final_loss"], "graph", d["config"].get("hip_graph"), "roofline", r.get("kernel"), r.get("shape"), r.get("frac"))
except Exception as e:
    print("$1 FAILED", e)
print(open("gpurun_out/bench_$1.err").read()[-600:])
PY
}
run graph_fusedbn "COT_FUSED_BN=1" ""
run graph_torchbn "COT_FUSED_BN=0" "--no-cpu-baseline"
run eager_fusedbn "COT_FUSED_BN=1" "--no-graph --no-cpu-baseline"
mkdir -p /tmp/prof && cd /tmp/prof && COT_ROCTX=1 timeout 400 rocprofv3 --kernel-trace --marker-trace --stats --output-format csv -d /tmp/prof/out -o trace -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 3 --no-cpu-baseline --no-kernel-timing > $GRAFT_REPO_ROOT/gpurun_out/prof_train.log 2>&1
cd $GRAFT_REPO_ROOT
tail -2 gpurun_out/prof_train.log | cut -c1-300
for f in $(find /tmp/prof/out -name "*stats*.csv"); do cp $f gpurun_out/; done
head -12 gpurun_out/trace_kernel_stats.csv | cut -c1-150
du -sh gpurun_out

#!/bin/bash
# forward-only (BASELINE config 2): eager with / without the eval single-call blocks, and the graph replay
mkdir -p gpurun_out; export TMPDIR=/tmp; O=gpurun_out
timeout 600 python -m pytest tests/test_fused_layer_gpu.py -m gpu -q -k "eval_mode" --timeout 300 -rfE -p no:cacheprovider --tb=short > $O/r05_fwd_pytest.log 2>&1; tail -3 $O/r05_fwd_pytest.log
for rep in 1 2; do
for t in "modules:COT_FUSED_LAYER=0:--mode fwd" "single:COT_X=1:--mode fwd" "graph:COT_X=1:--mode fwd --graph"; do
  name=${t%%:*}; rest=${t#*:}; envs=${rest%%:*}; flags=${rest#*:}
  ( for kv in ${envs//,/ }; do export $kv; done
    timeout 300 python bench.py --kernels new $flags --steps 50 --warmup 6 --settle-seconds 6 --no-cpu-baseline --no-kernel-timing --no-secondary --no-pmc > $O/r05_fwd_${name}_$rep.json 2> $O/r05_fwd_${name}_$rep.err || tail -4 $O/r05_fwd_${name}_$rep.err )
  python -c "
import json
d=json.load(open('$O/r05_fwd_${name}_$rep.json')); print('$name rep$rep', d['value'], d['ms_per_step'], d['host_issue_ms_per_step'], d['config']['nodes_per_step'])"
done; done

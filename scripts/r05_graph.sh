#!/bin/bash
# eager vs graph vs the N > 1 code path on one GPU (world-of-one RCCL), alternating on a warmed box
mkdir -p gpurun_out; export TMPDIR=/tmp; O=gpurun_out
timeout 200 python bench.py --kernels new --steps 150 --warmup 6 --settle-seconds 20 --no-cpu-baseline --no-kernel-timing --no-secondary --no-pmc > $O/r05_g_warm.json 2> $O/r05_g_warm.err
for rep in 1 2; do
for t in "eager:" "graph:--graph" "coll:--force-collectives" "collgraph:--force-collectives --graph"; do
  name=${t%%:*}; flags=${t#*:}
  timeout 300 python bench.py --kernels new $flags --steps 30 --warmup 6 --settle-seconds 8 --no-cpu-baseline --no-kernel-timing --no-secondary --no-pmc > $O/r05_g_${name}_$rep.json 2> $O/r05_g_${name}_$rep.err || tail -5 $O/r05_g_${name}_$rep.err
  python -c "
import json
d=json.load(open('$O/r05_g_${name}_$rep.json')); print('$name rep$rep', d['value'], d['ms_per_step'], d['final_loss'], d['host_issue_ms_per_step'], d['config']['nodes_per_step'], d['config']['grad_sync'])"
done; done

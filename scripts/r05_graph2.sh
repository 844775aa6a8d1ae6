#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp; O=gpurun_out
timeout 200 python bench.py --kernels new --steps 150 --warmup 6 --settle-seconds 20 --no-cpu-baseline --no-kernel-timing --no-secondary --no-pmc > $O/r05_g2_warm.json 2> $O/r05_g2_warm.err
for rep in 1 2; do
for t in "eager:" "collparam:--force-collectives --grad-dtype param" "collparamgraph:--force-collectives --grad-dtype param --graph" "collfp32:--force-collectives --grad-dtype fp32"; do
  name=${t%%:*}; flags=${t#*:}
  timeout 300 python bench.py --kernels new $flags --steps 30 --warmup 6 --settle-seconds 8 --no-cpu-baseline --no-kernel-timing --no-secondary --no-pmc 2> $O/r05_g2_${name}_$rep.err | grep '^{"metric' > $O/r05_g2_${name}_$rep.json || tail -5 $O/r05_g2_${name}_$rep.err
  python -c "
import json
d=json.load(open('$O/r05_g2_${name}_$rep.json')); print('$name rep$rep', d['value'], d['ms_per_step'], d['final_loss'], d['host_issue_ms_per_step'], d['config']['grad_sync'])"
done; done

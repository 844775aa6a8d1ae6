#!/bin/bash
# grouped 3x3, 16-channel groups on row tiles: odd channel-row stride (tuning key 49): per layer, then the whole step
mkdir -p gpurun_out; export TMPDIR=/tmp; O=gpurun_out
for v in 0 1 0 1; do COT_TUNING=49=$v timeout 300 python scripts/bench_conv_abi.py --iters 30 --only g4 2>&1 | grep -E "g4 .* 1  " | sed "s/^/49=$v /"; done | tee $O/slodd_abi.log | cut -c1-150
for rep in 1 2; do
for t in "even:49=0" "odd:49=1"; do
  name=${t%%:*}; tune=${t#*:}
  COT_TUNING=$tune timeout 300 python bench.py --kernels new --steps 20 --warmup 6 --settle-seconds 5 --no-cpu-baseline --no-kernel-timing --no-secondary --no-pmc > $O/slodd_step_${name}_$rep.json 2> $O/slodd_step_${name}_$rep.err || tail -4 $O/slodd_step_${name}_$rep.err
  python -c "
import json
d=json.load(open('$O/slodd_step_${name}_$rep.json')); print('$name rep$rep', d['value'], d['ms_per_step'], d['final_loss'])"
done; done | tee $O/slodd_step.log

#!/bin/bash
# weight gradients beside the backward pass: target workgroups per CU of the 1x1 weight-gradient GEMM (tuning key 25 bits 16..23 = x4; default 4 = one per CU)
mkdir -p gpurun_out; export TMPDIR=/tmp; O=gpurun_out
for rep in 1 2; do
for t in "one:0" "quarter:65536" "half:131072" "threeq:196608" "two:524288"; do
  name=${t%%:*}; tune=${t#*:}
  COT_TUNING=25=$tune timeout 300 python bench.py --kernels new --steps 20 --warmup 6 --settle-seconds 5 --no-cpu-baseline --no-kernel-timing --no-secondary --no-pmc > $O/wgcu_step_${name}_$rep.json 2> $O/wgcu_step_${name}_$rep.err || tail -4 $O/wgcu_step_${name}_$rep.err
  python -c "
import json
d=json.load(open('$O/wgcu_step_${name}_$rep.json')); print('$name rep$rep', d['value'], d['ms_per_step'], d['final_loss'])"
done; done | tee $O/wgcu_step.log

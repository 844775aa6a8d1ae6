#!/bin/bash
# after the LDS-layout fixes: fragment prefetch in the BIG tiles (23=4) and the few-tiles rule's threshold (46) again
mkdir -p gpurun_out; export TMPDIR=/tmp; O=gpurun_out
timeout 300 python scripts/probe_cnhw.py 30 23=4+46=200 23=0+46=400 23=0+46=100 23=0+46=200 2>&1 | grep -v amdgpu.ids | sed -n 1,62p > $O/post_probe.log; cat $O/post_probe.log | cut -c1-120
for rep in 1 2; do
for t in "base:" "pf:23=4" "fill400:46=400" "fill100:46=100"; do
  name=${t%%:*}; tune=${t#*:}
  COT_TUNING=$tune timeout 300 python bench.py --kernels new --steps 20 --warmup 6 --settle-seconds 5 --no-cpu-baseline --no-kernel-timing --no-secondary --no-pmc > $O/post_step_${name}_$rep.json 2> $O/post_step_${name}_$rep.err || tail -4 $O/post_step_${name}_$rep.err
  python -c "
import json
d=json.load(open('$O/post_step_${name}_$rep.json')); print('$name rep$rep', d['value'], d['ms_per_step'], d['final_loss'])"
done; done | tee $O/post_step.log

#!/bin/bash
# fragment prefetch in the BIG-tile 1x1 kernel (tuning key 23 bit 2): per-layer probe, then whole-step A/B
mkdir -p gpurun_out; export TMPDIR=/tmp; O=gpurun_out
timeout 300 python scripts/probe_cnhw.py 30 23=4 23=0 2>&1 | grep -v amdgpu.ids | sed -n 1,40p > $O/pf_probe.log; cat $O/pf_probe.log | cut -c1-120
for rep in 1 2; do
for t in "base:" "pf:23=4"; do
  name=${t%%:*}; tune=${t#*:}
  COT_TUNING=$tune timeout 300 python bench.py --kernels new --steps 20 --warmup 6 --settle-seconds 5 --no-cpu-baseline --no-kernel-timing --no-secondary --no-pmc > $O/pf_step_${name}_$rep.json 2> $O/pf_step_${name}_$rep.err || tail -4 $O/pf_step_${name}_$rep.err
  python -c "
import json
d=json.load(open('$O/pf_step_${name}_$rep.json')); print('$name rep$rep', d['value'], d['ms_per_step'], d['final_loss'])"
done; done

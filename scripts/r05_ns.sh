#!/bin/bash
# bank-conflict-free X stage of the BIG-tile 1x1 kernel (tuning key 48): per-layer, then whole-step A/B
mkdir -p gpurun_out; export TMPDIR=/tmp; O=gpurun_out
timeout 300 python scripts/probe_cnhw.py 30 48=0 48=1 2>&1 | grep -v amdgpu.ids | sed -n 1,40p > $O/xsw_probe.log; cat $O/xsw_probe.log | cut -c1-120
for v in 0 1; do COT_TUNING=48=$v timeout 300 python scripts/bench_conv_abi.py --iters 20 2>&1 | grep -E " 1  " | sed -n 1,10p > $O/xsw_abi_$v.log; done
paste -d'|' $O/xsw_abi_0.log $O/xsw_abi_1.log | cut -c1-230
for rep in 1 2; do
for t in "off:48=0" "on:48=1"; do
  name=${t%%:*}; tune=${t#*:}
  COT_TUNING=$tune timeout 300 python bench.py --kernels new --steps 20 --warmup 6 --settle-seconds 5 --no-cpu-baseline --no-kernel-timing --no-secondary --no-pmc > $O/xsw_step_${name}_$rep.json 2> $O/xsw_step_${name}_$rep.err || tail -4 $O/xsw_step_${name}_$rep.err
  python -c "
import json
d=json.load(open('$O/xsw_step_${name}_$rep.json')); print('$name rep$rep', d['value'], d['ms_per_step'], d['final_loss'])"
done; done

#!/bin/bash
# LDS bank conflicts of the 1x1 / grouped 3x3 kernels (tuning key 48: bit 0 X stage of BIG tiles, bit 1 W-tile permutation): per layer, then the whole step
mkdir -p gpurun_out; export TMPDIR=/tmp; O=gpurun_out
timeout 300 python scripts/probe_cnhw.py 30 48=3 48=7 2>&1 | grep -v amdgpu.ids | sed -n 1,40p > $O/xsw3_probe.log; cat $O/xsw3_probe.log | cut -c1-120
for v in 3 7; do COT_TUNING=48=$v timeout 300 python scripts/bench_conv_abi.py --iters 20 2>&1 | grep -E " 1  " > $O/xsw3_abi_$v.log; done
paste -d'|' $O/xsw3_abi_3.log $O/xsw3_abi_7.log | cut -c1-230
for rep in 1 2; do
for t in "xw:48=3" "xwt:48=7"; do
  name=${t%%:*}; tune=${t#*:}
  COT_TUNING=$tune timeout 300 python bench.py --kernels new --steps 20 --warmup 6 --settle-seconds 5 --no-cpu-baseline --no-kernel-timing --no-secondary --no-pmc > $O/xsw3_step_${name}_$rep.json 2> $O/xsw3_step_${name}_$rep.err || tail -4 $O/xsw3_step_${name}_$rep.err
  python -c "
import json
d=json.load(open('$O/xsw3_step_${name}_$rep.json')); print('$name rep$rep', d['value'], d['ms_per_step'], d['final_loss'])"
done; done

#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp; O=gpurun_out
for t in "25=0" "25=32" "25=34" "25=2"; do
  echo "== tune $t" >> $O/r3s14_wgrad_ab.log
  timeout 300 python scripts/bench_conv_abi.py --modes 1 --tune "$t" 2>&1 | grep "^s[0-9e]" | awk -F'|' '{print substr($1,1,30) "|" $3}' >> $O/r3s14_wgrad_ab.log
done
for v in "" "25=32"; do
  timeout 300 python bench.py --kernels new --steps 20 --warmup 6 --no-cpu-baseline --no-kernel-timing ${v:+--tune $v} > $O/r3s14_step.json 2> $O/r3s14_step.err || tail -3 $O/r3s14_step.err
  python -c "
import json
d=json.load(open('$O/r3s14_step.json')); print('STEP tune=$v', d['value'], d['ms_per_step'])"
done

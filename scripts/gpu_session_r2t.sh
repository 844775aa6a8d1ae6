#!/bin/bash
# A/B of the weight-gradient split heuristics on the headline step (cot_set_tuning 19: partial-sum bytes as % of input bytes; 11: target waves)
mkdir -p gpurun_out
export TMPDIR=/tmp
O=gpurun_out
T0=$(date +%s)
for t in "base:" "cap50:--tune 19=50" "cap100:--tune 19=100" "cap200:--tune 19=200" "cap100w4k:--tune 19=100,11=4096" "cap400:--tune 19=400" "base2:"; do
  tag=${t%%:*}; flags=${t#*:}
  timeout 300 python bench.py --kernels new $flags --steps 20 --warmup 6 --no-cpu-baseline --no-kernel-timing > $O/r2t_step_$tag.json 2> $O/r2t_step_$tag.err || tail -3 $O/r2t_step_$tag.err
  python -c "
import json
d=json.load(open('$O/r2t_step_$tag.json')); print('$tag', d['value'], d['ms_per_step'], d['final_loss'])"
done
echo "session wall=$(( $(date +%s) - T0 ))s"

#!/bin/bash
# general (fp32 / grouped) convolution kernels: first run on the device -- parity tests, per-op times next to MIOpen, and the
# bench lines of the fp32 recipe and of BASELINE configs 4 / 5 with both kernel sets
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
O=gpurun_out
T0=$(date +%s)
timeout 900 python -m pytest tests/test_conv_general_gpu.py tests/test_layers_gpu.py tests/test_conv1x1_gpu.py tests/test_group_norm9_gpu.py tests/test_conv3x3g_gpu.py -m gpu -q --timeout 300 -rfE -p no:cacheprovider > $O/gen_pytest.log 2>&1; echo "pytest rc=$? wall=$(( $(date +%s) - T0 ))s" >> $O/gen_pytest.log
grep -E "passed|failed|^FAILED|^ERROR|rc=" $O/gen_pytest.log | cut -c1-300 | tail -25
timeout 300 python scripts/bench_conv_general.py --json $O/gen_conv_general.json > $O/gen_conv_general.log 2>&1; cat $O/gen_conv_general.log | cut -c1-200
for cfg in "fp32:--dtype fp32" "cotnext101:--model cotnext101_2x48d --batch 64" "secotnetd152:--model se_cotnetd_152_L --img 320 --batch 64"; do
  tag=${cfg%%:*}; flags=${cfg#*:}
  for ks in round1 new; do
    timeout 400 python bench.py $flags --kernels $ks --steps 10 --warmup 3 --no-cpu-baseline > $O/gen_bench_${tag}_${ks}.json 2> $O/gen_bench_${tag}_${ks}.err || tail -5 $O/gen_bench_${tag}_${ks}.err
    python -c "
import json,sys
try:
    d=json.load(open('$O/gen_bench_${tag}_${ks}.json')); print('$tag $ks', d['value'], d['unit'], d['ms_per_step'], 'ms', d['final_loss'])
except Exception as e: print('$tag $ks: no line', e)"
  done
done
echo "session wall=$(( $(date +%s) - T0 ))s"

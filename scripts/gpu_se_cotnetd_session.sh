#!/bin/bash
# SE-CoTNetD layers on HIP (BlurPool, SplitAttn gate): parity, then the config 4 / 5 / fp32 lines on the final build
mkdir -p gpurun_out
export TMPDIR=/tmp
O=gpurun_out
timeout 600 python -m pytest tests/test_pool_gpu.py tests/test_se_gate_gpu.py tests/test_layers_gpu.py tests/test_radix_tail_gpu.py -m gpu -q --timeout 300 -rfE -p no:cacheprovider > $O/se_pytest.log 2>&1; echo "pytest rc=$?" >> $O/se_pytest.log
grep -E "passed|failed|^FAILED|^ERROR|rc=" $O/se_pytest.log | cut -c1-300 | tail -10
for cfg in "secotnetd152:--model se_cotnetd_152_L --img 320 --batch 64" "cotnext101:--model cotnext101_2x48d --batch 64" "fp32:--dtype fp32"; do
  tag=${cfg%%:*}; flags=${cfg#*:}
  for ks in new round1; do
    timeout 400 python bench.py $flags --kernels $ks --steps 10 --warmup 3 --no-cpu-baseline > $O/se_bench_${tag}_${ks}.json 2> $O/se_bench_${tag}_${ks}.err || tail -5 $O/se_bench_${tag}_${ks}.err
    python -c "
import json
try:
    d=json.load(open('$O/se_bench_${tag}_${ks}.json')); print('$tag $ks', d['value'], d['unit'], d['ms_per_step'], 'ms', d['final_loss'])
except Exception as e: print('$tag $ks: no line', e)"
  done
done

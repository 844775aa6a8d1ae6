#!/bin/bash
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
O=gpurun_out
T0=$(date +%s)
timeout 600 python -m pytest tests/test_fused_bn_gpu.py tests/test_conv_general_gpu.py tests/test_conv3x3g_gpu.py -m gpu -q --timeout 300 -rfE -p no:cacheprovider > $O/r2s_pytest.log 2>&1; echo "pytest rc=$?" >> $O/r2s_pytest.log
grep -E "passed|failed|^FAILED|^ERROR|rc=" $O/r2s_pytest.log | cut -c1-300 | tail -10
for t in "fold:" "nofold:--tune 12=0" "fold2:" "nofold2:--tune 12=0"; do
  tag=${t%%:*}; flags=${t#*:}
  timeout 300 python bench.py --kernels new $flags --steps 20 --warmup 6 --no-cpu-baseline --no-kernel-timing > $O/r2s_step_$tag.json 2> $O/r2s_step_$tag.err || tail -3 $O/r2s_step_$tag.err
  python -c "
import json
d=json.load(open('$O/r2s_step_$tag.json')); print('$tag', d['value'], d['ms_per_step'], d['final_loss'])"
done
timeout 300 python bench.py --kernels new --steps 20 --warmup 6 --no-cpu-baseline > $O/r2s_step_timing.json 2> $O/r2s_step_timing.err; cut -c1-150 $O/r2s_step_timing.json
echo "session wall=$(( $(date +%s) - T0 ))s"

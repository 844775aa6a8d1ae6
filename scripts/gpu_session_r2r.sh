#!/bin/bash
# re-test the general kernels after the weight-gradient split change; A/B of the folded BatchNorm finalize on the headline step
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
O=gpurun_out
T0=$(date +%s)
timeout 600 python -m pytest tests/test_conv_general_gpu.py -m gpu -q --timeout 300 -rfE -p no:cacheprovider > $O/r2r_pytest.log 2>&1; echo "pytest rc=$?" >> $O/r2r_pytest.log
grep -E "passed|failed|^FAILED|^ERROR|rc=" $O/r2r_pytest.log | cut -c1-300 | tail -10
timeout 300 python scripts/bench_conv_general.py --json $O/r2r_conv_general.json > $O/r2r_conv_general.log 2>&1; cat $O/r2r_conv_general.log | cut -c1-200
for t in "base:" "fold:--tune 12=1" "base2:" "fold2:--tune 12=1"; do
  tag=${t%%:*}; flags=${t#*:}
  timeout 300 python bench.py --kernels new $flags --steps 20 --warmup 6 --no-cpu-baseline --no-kernel-timing > $O/r2r_step_$tag.json 2> $O/r2r_step_$tag.err || tail -3 $O/r2r_step_$tag.err
  python -c "
import json
d=json.load(open('$O/r2r_step_$tag.json')); print('$tag', d['value'], d['ms_per_step'], d['final_loss'])"
done
timeout 300 python bench.py --dtype fp32 --kernels new --steps 10 --warmup 3 --no-cpu-baseline > $O/r2r_bench_fp32_new.json 2> $O/r2r_bench_fp32_new.err; cut -c1-200 $O/r2r_bench_fp32_new.json
echo "session wall=$(( $(date +%s) - T0 ))s"

#!/bin/bash
# byte-tap max pooling on the device; A/B of the LDS weight-gradient split cap (key 20)
mkdir -p gpurun_out
export TMPDIR=/tmp
O=gpurun_out
timeout 600 python -m pytest tests/test_pool_gpu.py tests/test_stem_gpu.py -m gpu -q --timeout 300 -rfE -p no:cacheprovider > $O/r2u_pytest.log 2>&1; echo "pytest rc=$?" >> $O/r2u_pytest.log
grep -E "passed|failed|^FAILED|^ERROR|rc=" $O/r2u_pytest.log | cut -c1-300 | tail -10
for t in "base:" "lds50:--tune 20=50" "lds100:--tune 20=100" "lds200:--tune 20=200" "base2:"; do
  tag=${t%%:*}; flags=${t#*:}
  timeout 300 python bench.py --kernels new $flags --steps 20 --warmup 6 --no-cpu-baseline --no-kernel-timing > $O/r2u_step_$tag.json 2> $O/r2u_step_$tag.err || tail -3 $O/r2u_step_$tag.err
  python -c "
import json
d=json.load(open('$O/r2u_step_$tag.json')); print('$tag', d['value'], d['ms_per_step'], d['final_loss'])"
done
bash scripts/gpu_trace_new.sh r2u_trace > $O/r2u_trace_sh.log 2>&1; tail -2 $O/r2u_trace_sh.log | cut -c1-200

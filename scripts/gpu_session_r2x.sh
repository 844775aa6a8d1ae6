#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
O=gpurun_out
timeout 600 python -m pytest tests/test_fused_bn_gpu.py tests/test_conv1x1_gpu.py tests/test_fused_layer_gpu.py tests/test_pool_gpu.py -m gpu -q --timeout 300 -rfE -p no:cacheprovider > $O/r2x_pytest.log 2>&1; echo "pytest rc=$?" >> $O/r2x_pytest.log
grep -E "passed|failed|^FAILED|^ERROR|rc=" $O/r2x_pytest.log | cut -c1-300 | tail -10
for t in "dflt:" "nochan:--tune 21=0" "nogenwg:--tune 17=16" "dflt2:"; do
  tag=${t%%:*}; flags=${t#*:}
  timeout 300 python bench.py --kernels new $flags --steps 20 --warmup 6 --no-cpu-baseline --no-kernel-timing > $O/r2x_step_$tag.json 2> $O/r2x_step_$tag.err || tail -3 $O/r2x_step_$tag.err
  python -c "
import json
d=json.load(open('$O/r2x_step_$tag.json')); print('$tag', d['value'], d['ms_per_step'], d['final_loss'])"
done

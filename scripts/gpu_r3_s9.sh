#!/bin/bash
# round 3, session 9: weight gradients on a side stream: parity, then whole-step A/B
mkdir -p gpurun_out; export TMPDIR=/tmp; O=gpurun_out
timeout 900 python -m pytest tests/test_fused_layer_gpu.py tests/test_layers_gpu.py tests/test_conv1x1_gpu.py -m gpu -q -x --timeout 300 -p no:cacheprovider > $O/r3s9_pytest.log 2>&1; echo "pytest rc=$?" >> $O/r3s9_pytest.log
tail -3 $O/r3s9_pytest.log
for v in 1 0 1 0; do
  COT_WGRAD_STREAM=$v timeout 300 python bench.py --kernels new --steps 20 --warmup 6 --no-cpu-baseline --no-kernel-timing > $O/r3s9_step_$v.json 2> $O/r3s9_step_$v.err || tail -3 $O/r3s9_step_$v.err
  python -c "
import json
d=json.load(open('$O/r3s9_step_$v.json')); print('side stream $v:', d['value'], d['ms_per_step'], d['final_loss'])"
done

#!/bin/bash
# round 3, session 10: stochastic depth / recipe on the single-node path: parity, then the recipe step beside the plain one
mkdir -p gpurun_out; export TMPDIR=/tmp; O=gpurun_out
timeout 900 python -m pytest tests/test_fused_layer_gpu.py tests/test_fused_bn_gpu.py tests/test_head_gpu.py -m gpu -q -x --timeout 300 -p no:cacheprovider > $O/r3s10_pytest.log 2>&1; echo "pytest rc=$?" >> $O/r3s10_pytest.log
tail -4 $O/r3s10_pytest.log
for v in "" "--recipe" "" "--recipe"; do
  n=plain; [ -n "$v" ] && n=recipe
  timeout 300 python bench.py --kernels new $v --steps 20 --warmup 6 --no-cpu-baseline --no-kernel-timing > $O/r3s10_step_$n.json 2> $O/r3s10_step_$n.err || tail -3 $O/r3s10_step_$n.err
  python -c "
import json
d=json.load(open('$O/r3s10_step_$n.json')); print('$n:', d['value'], d['ms_per_step'], d['final_loss'])"
done

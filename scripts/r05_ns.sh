#!/bin/bash
# radix-2 tail on 7 x 7 planes: eight planes per wave (tuning key 50): per kernel, GPU tests, then the whole step
mkdir -p gpurun_out; export TMPDIR=/tmp; O=gpurun_out
timeout 200 python scripts/bench_radix.py 50 50=0 50=1 50=0 50=1 2>&1 | grep -v amdgpu.ids | tee $O/radix7_layers.log
timeout 400 python -m pytest tests/test_radix_tail_gpu.py tests/test_step_kernels_b80_gpu.py tests/test_layouts_gpu.py tests/test_fused_layer_gpu.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -3
for rep in 1 2; do
for t in "one:50=0" "packed:50=1"; do
  name=${t%%:*}; tune=${t#*:}
  COT_TUNING=$tune timeout 300 python bench.py --kernels new --steps 20 --warmup 6 --settle-seconds 5 --no-cpu-baseline --no-kernel-timing --no-secondary --no-pmc > $O/radix7_step_${name}_$rep.json 2> $O/radix7_step_${name}_$rep.err || tail -4 $O/radix7_step_${name}_$rep.err
  python -c "
import json
d=json.load(open('$O/radix7_step_${name}_$rep.json')); print('$name rep$rep', d['value'], d['ms_per_step'], d['final_loss'])"
done; done | tee $O/radix7_step.log

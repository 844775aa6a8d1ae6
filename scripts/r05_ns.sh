#!/bin/bash
# GroupNorm-9 on small planes: several (image, group) pairs per wave (tuning key 49): per layer, then the whole step
mkdir -p gpurun_out; export TMPDIR=/tmp; O=gpurun_out
timeout 200 python scripts/bench_gn9.py 50 49=0 49=1 49=0 49=1 2>&1 | grep -v amdgpu.ids | tee $O/gn9pack_layers.log
timeout 300 python -m pytest tests/test_layers_gpu.py tests/test_layouts_gpu.py tests/test_fused_layer_gpu.py -m gpu -q -x -p no:cacheprovider -k "group_norm or gn or cm or channel_major or stage" 2>&1 | tail -3
for rep in 1 2; do
for t in "one:49=0" "packed:49=1"; do
  name=${t%%:*}; tune=${t#*:}
  COT_TUNING=$tune timeout 300 python bench.py --kernels new --steps 20 --warmup 6 --settle-seconds 5 --no-cpu-baseline --no-kernel-timing --no-secondary --no-pmc > $O/gn9pack_step_${name}_$rep.json 2> $O/gn9pack_step_${name}_$rep.err || tail -4 $O/gn9pack_step_${name}_$rep.err
  python -c "
import json
d=json.load(open('$O/gn9pack_step_${name}_$rep.json')); print('$name rep$rep', d['value'], d['ms_per_step'], d['final_loss'])"
done; done | tee $O/gn9pack_step.log

#!/bin/bash
# K tail of the LDS 1x1 kernels (tuning key 54): tests, then alternating A/B on the default line and on CoTNeXt-101 / -50
mkdir -p gpurun_out; export TMPDIR=/tmp; O=gpurun_out
timeout 1200 python -m pytest tests/test_conv1x1_gpu.py tests/test_conv_general_gpu.py tests/test_fused_layer_gpu.py tests/test_layouts_gpu.py tests/test_dispatch_parity_gpu.py tests/test_fuzz_gpu.py -x -q > $O/r06_ktail_pytest.log 2>&1; tail -4 $O/r06_ktail_pytest.log
run() { python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing --no-secondary --no-pmc $2 --tune "$1" 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('%-12s %-40s %.3f ms/step %.1f img/s' % ('$1', '$2', d['ms_per_step'], d['value']))"; }
for i in 1 2 3; do run 54=0 ""; run 54=1 ""; done | tee $O/r06_ktail_ab.log
for i in 1 2 3; do run 54=0 "--model cotnext101_2x48d --batch 64"; run 54=1 "--model cotnext101_2x48d --batch 64"; done | tee -a $O/r06_ktail_ab.log
for i in 1 2; do run 54=0 "--model cotnext50_2x48d --batch 80"; run 54=1 "--model cotnext50_2x48d --batch 80"; done | tee -a $O/r06_ktail_ab.log

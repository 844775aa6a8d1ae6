#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp; O=gpurun_out
timeout 900 python -m pytest tests/test_fused_layer_gpu.py tests/test_layers_gpu.py -m gpu -x -q 2>&1 | grep -E "passed|failed" | tail -2 > $O/r3s38.log
for i in 1 2; do
COT_NO_PROBE_CACHE=1 timeout 400 python bench.py --kernels new --steps 30 --warmup 10 --no-cpu-baseline --no-kernel-timing 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('step', d['ms_per_step'])" >> $O/r3s38.log
done

#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp; O=gpurun_out
timeout 900 python -m pytest tests/test_fused_layer_gpu.py tests/test_flat_sgd_gpu.py tests/test_fused_bn_gpu.py tests/test_rccl_gpu.py -m gpu -x -q 2>&1 | tail -2 > $O/r3s36_ab.log
for i in 1 2 3; do
COT_WGRAD_LAZY=0 COT_NO_PROBE_CACHE=1 timeout 400 python bench.py --kernels new --steps 30 --warmup 10 --no-cpu-baseline --no-kernel-timing 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('per-launch waits', d['ms_per_step'])" >> $O/r3s36_ab.log
COT_NO_PROBE_CACHE=1 timeout 400 python bench.py --kernels new --steps 30 --warmup 10 --no-cpu-baseline --no-kernel-timing 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('lazy flush      ', d['ms_per_step'])" >> $O/r3s36_ab.log
done

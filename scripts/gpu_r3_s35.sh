#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp; O=gpurun_out
timeout 600 python -m pytest tests/test_fused_layer_gpu.py -m gpu -x -q 2>&1 | tail -2 > $O/r3s35_ab.log
for i in 1 2 3; do
COT_FWD_OVERLAP=0 COT_NO_PROBE_CACHE=1 timeout 400 python bench.py --kernels new --steps 30 --warmup 10 --no-cpu-baseline --no-kernel-timing 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('serial ', d['ms_per_step'])" >> $O/r3s35_ab.log
COT_NO_PROBE_CACHE=1 timeout 400 python bench.py --kernels new --steps 30 --warmup 10 --no-cpu-baseline --no-kernel-timing 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('overlap', d['ms_per_step'])" >> $O/r3s35_ab.log
done

#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp; O=gpurun_out
for b in 80 40 20 80; do
COT_NO_PROBE_CACHE=1 timeout 400 python bench.py --kernels new --steps 30 --warmup 10 --no-cpu-baseline --no-kernel-timing --batch $b 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('batch $b', d['ms_per_step'], d['value'], d['settle'])" >> $O/r3s32_batch.log
done

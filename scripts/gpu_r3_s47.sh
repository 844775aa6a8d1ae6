#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp; O=gpurun_out
timeout 200 python -m pytest tests/test_fused_layer_gpu.py tests/test_flat_sgd_gpu.py -m gpu -q -p no:cacheprovider 2>&1 | grep -E "passed|failed|FAILED" | tail -3 > $O/r3s47.log
COT_NO_PROBE_CACHE=1 timeout 200 python bench.py --kernels new --steps 30 --warmup 10 --no-cpu-baseline --no-kernel-timing 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('step', d['ms_per_step'], d['value'])" >> $O/r3s47.log

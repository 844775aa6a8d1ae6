#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp; O=gpurun_out
timeout 300 python -m pytest tests/test_fused_bn_gpu.py tests/test_fused_layer_gpu.py -m gpu -q -p no:cacheprovider 2>&1 | grep -E "passed|failed|FAILED" | tail -3 > $O/r3s46.log
for t in "28=2048" "28=1024" "28=2048" "28=1024"; do
COT_NO_PROBE_CACHE=1 timeout 300 python bench.py --kernels new --steps 30 --warmup 10 --no-cpu-baseline --no-kernel-timing --tune $t 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$t', d['ms_per_step'])" >> $O/r3s46.log
done

#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp; O=gpurun_out
for v in 0 2 64 128; do timeout 200 python scripts/check_wgrad_variant.py $v 2>&1 | tail -1 >> $O/r3s26_check.log; done
timeout 600 python -m pytest tests/test_conv1x1_gpu.py tests/test_fused_layer_gpu.py -m gpu -x -q 2>&1 | tail -2 >> $O/r3s26_check.log
for t in "25=64" "25=0" "25=64" "25=0"; do
  echo "== tune $t" >> $O/r3s26_wgrad_ab.log
  timeout 300 python scripts/bench_conv_abi.py --modes 1 --tune "$t" 2>&1 | grep "^s[0-9e]" | awk -F'|' '{print substr($1,1,30) "|" $3}' >> $O/r3s26_wgrad_ab.log
done
COT_NO_PROBE_CACHE=1 timeout 400 python bench.py --kernels new --steps 30 --warmup 10 --tune 25=64 > $O/r3s26_bench_k32.json 2> $O/r3s26_bench_k32.err
COT_NO_PROBE_CACHE=1 timeout 400 python bench.py --kernels new --steps 30 --warmup 10 > $O/r3s26_bench_k64.json 2> $O/r3s26_bench_k64.err
COT_NO_PROBE_CACHE=1 timeout 400 python bench.py --kernels new --steps 30 --warmup 10 --tune 25=64 > $O/r3s26_bench_k32b.json 2> $O/r3s26_bench_k32b.err
COT_NO_PROBE_CACHE=1 timeout 400 python bench.py --kernels new --steps 30 --warmup 10 > $O/r3s26_bench_k64b.json 2> $O/r3s26_bench_k64b.err

#!/bin/bash
# scalar-light weight-gradient loop: parity, per-shape A/B (25=0 new PF loop, 25=2 PF off), whole-step bench
mkdir -p gpurun_out; export TMPDIR=/tmp; O=gpurun_out
timeout 600 python -m pytest tests/test_conv1x1_gpu.py tests/test_fused_layer_gpu.py -m gpu -x -q > $O/r3s17_tests.log 2>&1; tail -3 $O/r3s17_tests.log
for t in "25=0" "25=2" "25=0" "25=2"; do
  echo "== tune $t" >> $O/r3s17_wgrad_ab.log
  timeout 300 python scripts/bench_conv_abi.py --modes 1 --tune "$t" 2>&1 | grep "^s[0-9e]" | awk -F'|' '{print substr($1,1,30) "|" $3}' >> $O/r3s17_wgrad_ab.log
done
COT_NO_PROBE_CACHE=1 timeout 400 python bench.py --kernels new --steps 30 --warmup 10 > $O/r3s17_bench_new.json 2> $O/r3s17_bench_new.err
tail -c 600 $O/r3s17_bench_new.json

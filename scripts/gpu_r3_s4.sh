#!/bin/bash
# round 3, session 4: third-generation 1x1 weight gradient: parity on the device, then per-layer A/B of its knobs
mkdir -p gpurun_out; export TMPDIR=/tmp; O=gpurun_out
timeout 900 python -m pytest tests/test_conv1x1_gpu.py tests/test_fused_layer_gpu.py tests/test_layers_gpu.py -m gpu -q -x --timeout 300 -p no:cacheprovider > $O/r3s4_pytest.log 2>&1; echo "pytest rc=$?" >> $O/r3s4_pytest.log
tail -3 $O/r3s4_pytest.log
for t in "25=1" "25=0" "25=2" "25=4" "25=262144" "25=1048576" "25=6400" "25=25600"; do
  echo "== tune $t" >> $O/r3s4_wgrad_ab.log
  timeout 300 python scripts/bench_conv_abi.py --modes 1 --tune "$t" 2>&1 | grep "^s[0-9e]" >> $O/r3s4_wgrad_ab.log
done

#!/bin/bash
# round 3, session 2: third-generation 1x1 forward / data-gradient kernel: parity on the device, then per-layer A/B
mkdir -p gpurun_out; export TMPDIR=/tmp; O=gpurun_out
timeout 900 python -m pytest tests/test_conv1x1_gpu.py tests/test_fused_layer_gpu.py -m gpu -q -x --timeout 300 -p no:cacheprovider > $O/r3s2_pytest.log 2>&1; echo "pytest rc=$?" >> $O/r3s2_pytest.log
tail -3 $O/r3s2_pytest.log
for t in "23=1" "23=0" "23=2" "23=4" "23=6"; do
  echo "== tune $t" >> $O/r3s2_conv_ab.log
  timeout 300 python scripts/bench_conv_abi.py --modes 1 --tune "$t" 2>&1 | grep -v "grouped\|^C[0-9]\|amdgpu.ids" >> $O/r3s2_conv_ab.log
done

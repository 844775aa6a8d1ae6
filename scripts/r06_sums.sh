#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp; O=gpurun_out
timeout 900 python -m pytest tests/test_bn_tail_gpu.py tests/test_fused_layer_gpu.py tests/test_layouts_gpu.py -x -q > $O/r06_sums_pytest.log 2>&1; tail -4 $O/r06_sums_pytest.log
python scripts/bench_bn_tail.py 2>&1 | tee $O/r06_bn_tail_kernels5.log
bash scripts/r06_ab.sh "COT_BN_TAIL=0" "COT_BN_TAIL=1" 3 | tee $O/r06_bn_tail_ab2.log

#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp; O=gpurun_out
timeout 900 python -m pytest tests/test_bn_tail_gpu.py tests/test_fused_layer_gpu.py tests/test_layouts_gpu.py tests/test_conv1x1_gpu.py -x -q > $O/r06_resfold_pytest.log 2>&1; tail -4 $O/r06_resfold_pytest.log
bash scripts/r06_ab.sh "COT_RES_FOLD=0" "COT_RES_FOLD=1" 3 | tee $O/r06_res_fold_ab.log

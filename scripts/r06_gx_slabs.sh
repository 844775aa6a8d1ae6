#!/bin/bash
# CoXtLayer.embed[0] in channel-major blocks as two-slab 1x1 kernels per group (COT_GX_SLABS, cot_block_cm.py): tests, then A/B on CoTNeXt
mkdir -p gpurun_out; export TMPDIR=/tmp; O=gpurun_out
timeout 900 python -m pytest tests/test_fused_layer_gpu.py tests/test_layouts_gpu.py -x -q > $O/r06_gx_slabs_pytest.log 2>&1; tail -4 $O/r06_gx_slabs_pytest.log
bash scripts/r06_ab.sh "COT_GX_SLABS=0" "COT_GX_SLABS=1" 3 "--model cotnext101_2x48d --batch 64" | tee $O/r06_gx_slabs_ab.log
bash scripts/r06_ab.sh "COT_GX_SLABS=0" "COT_GX_SLABS=1" 2 "--model cotnext50_2x48d --batch 80" | tee -a $O/r06_gx_slabs_ab.log

#!/bin/bash
# CoXtLayer(96).key_embed (8 groups of 12 channels) as 4 groups of 24 with a block-diagonal weight on the LDS 3x3 kernels (COT_MERGE12)
mkdir -p gpurun_out; export TMPDIR=/tmp; O=gpurun_out
timeout 900 python -m pytest tests/test_fused_layer_gpu.py tests/test_layers_gpu.py tests/test_fuzz_nodes_gpu.py -x -q > $O/r06_merge12_pytest.log 2>&1; tail -4 $O/r06_merge12_pytest.log
bash scripts/r06_ab.sh "COT_MERGE12=0" "COT_MERGE12=1" 3 "--model cotnext101_2x48d --batch 64" | tee $O/r06_merge12_ab.log
bash scripts/r06_ab.sh "COT_MERGE12=0" "COT_MERGE12=1" 2 "--model cotnext50_2x48d --batch 80" | tee -a $O/r06_merge12_ab.log

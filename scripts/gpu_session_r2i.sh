#!/bin/bash
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
O=gpurun_out
timeout 400 python -m pytest tests/test_layers_gpu.py tests/test_fused_bn_gpu.py tests/test_fused_layer_gpu.py tests/test_rccl_gpu.py -m gpu -q --timeout 240 -rfE -p no:cacheprovider > $O/r2i_pytest.log 2>&1; echo "pytest rc=$?" >> $O/r2i_pytest.log
grep -E "passed|failed|^FAILED|^ERROR|rc=|^E  " $O/r2i_pytest.log | cut -c1-300 | tail -20
timeout 100 python scripts/diag_7x7.py > $O/r2i_diag_7x7.log 2>&1; head -8 $O/r2i_diag_7x7.log | cut -c1-220

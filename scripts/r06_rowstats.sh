#!/bin/bash
# GPU session: aggregation forward with bn's row statistics (COT_AGG_ROWSTATS) -- parity tests, op-level cost, alternating A/B of the default line
mkdir -p gpurun_out; export TMPDIR=/tmp; O=gpurun_out
timeout 900 python -m pytest tests/test_bn_tail_gpu.py tests/test_fused_layer_gpu.py tests/test_layouts_gpu.py tests/test_agg_gpu.py -x -q > $O/r06_rowstats_pytest.log 2>&1; tail -4 $O/r06_rowstats_pytest.log
python scripts/bench_rowstats.py 2>&1 | tee $O/r06_rowstats_kernels.log
bash scripts/r06_ab.sh "COT_AGG_ROWSTATS=0" "COT_AGG_ROWSTATS=1" 3 | tee $O/r06_rowstats_ab.log

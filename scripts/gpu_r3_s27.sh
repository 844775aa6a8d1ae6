#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp; O=gpurun_out
timeout 300 python scripts/bench_conv3x3g_wgrad.py > $O/r3s27_c3wgrad.log 2>&1
timeout 600 python -m pytest tests/test_conv3x3g_gpu.py tests/test_fused_layer_gpu.py -m gpu -x -q 2>&1 | tail -3 >> $O/r3s27_c3wgrad.log

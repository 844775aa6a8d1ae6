#!/bin/bash
set -x
mkdir -p gpurun_out
timeout 500 python scripts/bench_agg_abi.py --iters 20 --rounds 3 --variants v2,v2_fP8,v3,v3_fP8,v3_jp2,v3_jp8,v3_bP2 --out gpurun_out/agg_ab_v3.json > gpurun_out/agg_ab_v3.log 2>&1; grep -v "max|diff| out/gx/gw = \[0.0, 0.0, 0.0\]" gpurun_out/agg_ab_v3.log | cut -c1-260

#!/bin/bash
# (run at commit 5f0d3c1-era tree, before the NHWC study files it exercises were removed; kept as the record of how profiles/r05_channels_last_study_* were produced)
# round 5, session 1: the channels-last study route on hardware: its GPU tests, the GEMM/layer micro-benchmarks
# and a whole-step A/B of COT_CHANNELS_LAST_STUDY=1 against the default
mkdir -p gpurun_out; export TMPDIR=/tmp; O=gpurun_out
COT_STUDY_GPU=1 timeout 600 python -m pytest tests/test_channels_last_study_gpu.py -m gpu -q --timeout 200 -rfE -p no:cacheprovider --tb=short > $O/r05_cl_pytest.log 2>&1; echo "pytest rc=$?" >> $O/r05_cl_pytest.log
tail -40 $O/r05_cl_pytest.log | cut -c1-250
timeout 200 python scripts/bench_gemm_kc.py 30 > $O/r05_gemm_kc.log 2>&1; tail -12 $O/r05_gemm_kc.log | cut -c1-250
COT_PROFILE_ALL=1 timeout 300 python scripts/bench_cot_layer_channels_last.py 10 > $O/r05_cl_layer.log 2>&1; tail -30 $O/r05_cl_layer.log | cut -c1-250
for rep in 1 2; do
for v in 0 1; do
  COT_CHANNELS_LAST_STUDY=$v timeout 300 python bench.py --kernels new --steps 20 --warmup 6 --settle-seconds 5 --no-cpu-baseline --no-kernel-timing --no-secondary > $O/r05_step_cl${v}_$rep.json 2> $O/r05_step_cl${v}_$rep.err || tail -5 $O/r05_step_cl${v}_$rep.err
  python -c "
import json
d=json.load(open('$O/r05_step_cl${v}_$rep.json')); print('cl=$v rep=$rep', d['value'], d['ms_per_step'], d.get('final_loss'))"
done; done

#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp; O=gpurun_out
timeout 900 python -m pytest tests/test_stem_gpu.py tests/test_head_gpu.py tests/test_conv1x1_gpu.py tests/test_conv_general_gpu.py -x -q > $O/r06_fp32_pytest.log 2>&1; tail -5 $O/r06_fp32_pytest.log
python bench.py --dtype fp32 --batch 80 --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --no-pmc > $O/r06_bench_fp32.json 2> $O/r06_bench_fp32.err; python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r06_bench_fp32.json') if l.startswith('{')][-1])
print('fp32:', d['value'], 'img/s', d['ms_per_step'], 'ms/step', d['config'].get('module_fallbacks_per_step'), d['config'].get('nodes_per_step'))
PY
tail -3 $O/r06_bench_fp32.err

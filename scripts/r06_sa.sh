#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp; O=gpurun_out
timeout 900 python -m pytest tests/test_fused_layer_gpu.py tests/test_se_gate_gpu.py tests/test_layers_gpu.py -x -q > $O/r06_sa_pytest.log 2>&1; tail -4 $O/r06_sa_pytest.log
python bench.py --model se_cotnetd_152_L --img 320 --batch 64 --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --no-pmc --no-kernel-timing 2>/dev/null > $O/r06_bench_secotnetd152.json; python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r06_bench_secotnetd152.json') if l.startswith('{')][-1])
print('se_cotnetd_152_L:', d['value'], 'img/s', d['ms_per_step'], 'ms/step', d['config'].get('module_fallbacks_per_step'), d['config'].get('nodes_per_step'))
PY
bash scripts/gpu_trace_new.sh r06_secot --no-pmc --model se_cotnetd_152_L --img 320 --batch 64 > $O/r06_secot_trace_sh.log 2>&1
grep -c "Cijk" $O/r06_secot_per_shape.csv; grep "Cijk\|at::native::reduce" $O/r06_secot_per_shape.csv | cut -c1-120 | head -5

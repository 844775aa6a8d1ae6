#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp; O=gpurun_out
timeout 900 python -m pytest tests/test_bn_tail_gpu.py tests/test_fused_layer_gpu.py tests/test_layouts_gpu.py -x -q > $O/r06_se_pytest.log 2>&1; tail -4 $O/r06_se_pytest.log
bash scripts/r06_ab.sh "COT_SE_FUSED=0" "COT_SE_FUSED=1" 3 | tee $O/r06_se_fused_ab.log
bash scripts/r05_ksum.sh r06_se "off:COT_SE_FUSED=0:" "on:COT_SE_FUSED=1:"
python - <<'PY' | tee -a gpurun_out/r06_se_fused_ab.log
import json
a=json.load(open('gpurun_out/r06_se_ksum_off.json')); b=json.load(open('gpurun_out/r06_se_ksum_on.json'))
fam=lambda d:{r['kernel']:(r['ms_per_step'],r['launches_per_step'],r['avg_us']) for r in d['kernels']}
fa,fb=fam(a),fam(b)
print('kernel families that differ (ms per step, launches, avg us): COT_SE_FUSED=0 -> =1')
for k in sorted(set(fa)|set(fb), key=lambda k:-abs(fb.get(k,(0,0,0))[0]-fa.get(k,(0,0,0))[0])):
    x,y=fa.get(k,(0,0,0)),fb.get(k,(0,0,0))
    if abs(x[0]-y[0])>0.004: print('%-40s %7.3f (%4d, %5.1f us) -> %7.3f (%4d, %5.1f us)  %+.3f'%(k[:40],x[0],x[1],x[2],y[0],y[1],y[2],y[0]-x[0]))
print('sum of library kernels: %.3f -> %.3f ms'%(a['library_ms_per_step'],b['library_ms_per_step']))
PY

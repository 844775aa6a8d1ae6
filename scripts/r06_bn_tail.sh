#!/bin/bash
# GPU session for the BatchNorm + SiLU -> radix-tail fusion (COT_BN_TAIL): parity tests, alternating A/B of the default line, per-kernel families
mkdir -p gpurun_out; export TMPDIR=/tmp; O=gpurun_out
timeout 900 python -m pytest tests/test_bn_tail_gpu.py tests/test_radix_tail_gpu.py tests/test_fused_layer_gpu.py tests/test_layouts_gpu.py -x -q > $O/r06_bn_tail_pytest.log 2>&1; tail -5 $O/r06_bn_tail_pytest.log
bash scripts/r06_ab.sh "COT_BN_TAIL=0" "COT_BN_TAIL=1" 3 | tee $O/r06_bn_tail_ab.log
bash scripts/r05_ksum.sh r06_bn_tail "off:COT_BN_TAIL=0:" "on:COT_BN_TAIL=1:"
python - <<'PY' | tee -a gpurun_out/r06_bn_tail_ab.log
import json
a=json.load(open('gpurun_out/r06_bn_tail_ksum_off.json')); b=json.load(open('gpurun_out/r06_bn_tail_ksum_on.json'))
def fam(d):
    return {r['kernel']: (r['ms_per_step'], r['launches_per_step']) for r in d['kernels']}
fa,fb=fam(a),fam(b)
print('kernel families that differ (ms per step, launches): COT_BN_TAIL=0 -> =1')
ta=tb=0
for k in sorted(set(fa)|set(fb), key=lambda k:-abs(fb.get(k,(0,0))[0]-fa.get(k,(0,0))[0])):
    x,y=fa.get(k,(0,0)),fb.get(k,(0,0)); ta+=x[0]; tb+=y[0]
    if abs(x[0]-y[0])>0.004: print('%-44s %7.3f (%4d) -> %7.3f (%4d)  %+.3f'%(k[:44],x[0],x[1],y[0],y[1],y[0]-x[0]))
print('sum of library kernels: %.3f -> %.3f ms'%(ta,tb))
PY

#!/bin/bash
# round 3, session 13: the bench line with the widened roofline object (probe verdict cached after the first run of the session)
mkdir -p gpurun_out; export TMPDIR=/tmp; O=gpurun_out
timeout 900 python bench.py --no-cpu-baseline > $O/r3s13_bench.json 2> $O/r3s13_bench.err || tail -5 $O/r3s13_bench.err
python - <<'P'
import json
d=json.load(open('gpurun_out/r3s13_bench.json'))
print('STEP', d['value'], d['ms_per_step'])
r=d['roofline']
print(json.dumps(r['conv_bn_families'], indent=0))
for row in r['conv_bn_calls'][:40]:
    print(f"{row['op']:15s} {row['shape']:28s} x{row['calls_per_step']:<5} {row['avg_us']:8.1f} us {row['frac_hbm']*100:5.1f}% hbm  {row.get('TFLOPs','')!s:>6} TF  {row['ms_per_step']:.3f} ms  {','.join(row['kernels'])[:60]}")
for k in r['kernels'][:8]:
    print(k['kernel'], k['shape'], k['avg_us'], k['frac'])
P

#!/bin/bash
# round 3, session 15: the other BASELINE configurations under --kernels auto (probe verdict recorded on the line)
mkdir -p gpurun_out; export TMPDIR=/tmp; O=gpurun_out
run() { name=$1; shift; ( time timeout 900 python bench.py --no-cpu-baseline "$@" > $O/r3s15_$name.json 2> $O/r3s15_$name.err ) 2> $O/r3s15_$name.time
  python -c "
import json
d=json.load(open('$O/r3s15_$name.json')); ks=d['config']['kernel_selection']
print('$name', d['value'], d['ms_per_step'], 'chosen', ks.get('chosen'), {k:(v.get('parity'), v.get('ms_per_step'), v.get('worst_block_ratio_to_gate'), v.get('worst_block'), v.get('error')) for k,v in ks.get('probe',{}).items()}, ks.get('probe_error'))" || tail -5 $O/r3s15_$name.err
  grep real $O/r3s15_$name.time; }
run config2_fwd --mode fwd --steps 30 --warmup 8
run cotnext101 --model cotnext101_2x48d --batch 64 --steps 10 --warmup 4
run secotnetd152 --model se_cotnetd_152_L --img 320 --batch 64 --steps 10 --warmup 4
run cotnet50_recipe --recipe

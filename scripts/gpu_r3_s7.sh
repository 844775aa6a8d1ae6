#!/bin/bash
# round 3, session 7: whole step on the current build (bench.py --kernels new) + kernel trace per shape
mkdir -p gpurun_out; export TMPDIR=/tmp; O=gpurun_out
timeout 300 python bench.py --kernels new --steps 20 --warmup 6 --no-cpu-baseline --no-kernel-timing > $O/r3s7_step.json 2> $O/r3s7_step.err || tail -3 $O/r3s7_step.err
python -c "
import json
d=json.load(open('$O/r3s7_step.json')); print('STEP', d['value'], d['ms_per_step'], d['final_loss'])"
timeout 300 python bench.py --kernels new --steps 20 --warmup 6 --no-cpu-baseline --no-kernel-timing --tune 23=1,25=1 > $O/r3s7_step_old.json 2> $O/r3s7_step_old.err
python -c "
import json
d=json.load(open('$O/r3s7_step_old.json')); print('STEP(old kernels)', d['value'], d['ms_per_step'], d['final_loss'])"
bash scripts/gpu_trace_new.sh r3s7_trace > /dev/null 2>&1
head -45 $O/r3s7_trace_per_shape.csv | cut -c1-150

#!/bin/bash
# alternating A/B of the default bench line on one box: bash scripts/r06_ab.sh "<env A>" "<env B>" [reps] [extra bench args]
mkdir -p gpurun_out
A="$1"; B="$2"; R=${3:-3}; X="$4"
run() { env $1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing --no-secondary --no-pmc $X 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('%-40s %.3f ms/step %.1f img/s' % ('$1' or '(default)', d['ms_per_step'], d['value']))"; }
for i in $(seq $R); do run "$A"; run "$B"; done

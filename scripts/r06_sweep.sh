#!/bin/bash
# one-box sweep of library tuning keys on the default bench line: bash scripts/r06_sweep.sh "46=0" "46=300" ...   (COT_TUNING syntax)
mkdir -p gpurun_out
run() { COT_TUNING="$1" python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing --no-secondary --no-pmc 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('%-24s %.3f ms/step %.1f img/s' % ('$1' or '(default)', d['ms_per_step'], d['value']))"; }
for rep in 1 2; do run ""; for t in "$@"; do run "$t"; done; done

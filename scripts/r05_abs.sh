#!/bin/bash
# absolute step time after the box has warmed up: 40 s of load first, then alternate
mkdir -p gpurun_out; export TMPDIR=/tmp; O=gpurun_out
timeout 200 python bench.py --kernels new --steps 200 --warmup 6 --settle-seconds 30 --no-cpu-baseline --no-kernel-timing --no-secondary > $O/r05_abs_warm.json 2> $O/r05_abs_warm.err
python -c "
import json
d=json.load(open('$O/r05_abs_warm.json')); print('warm', d['value'], d['ms_per_step'], d['settle'])"
for rep in 1 2; do for v in 0 1; do
  COT_CM_LAYOUT=$v timeout 300 python bench.py --kernels new --steps 30 --warmup 6 --settle-seconds 10 --no-cpu-baseline --no-kernel-timing --no-secondary > $O/r05_abs_cm${v}_$rep.json 2> $O/r05_abs_cm${v}_$rep.err
  python -c "
import json
d=json.load(open('$O/r05_abs_cm${v}_$rep.json')); print('cm=$v rep$rep', d['value'], d['ms_per_step'], d['settle'], d['host_issue_ms_per_step'])"
done; done
for v in 0 1; do
COT_CM_LAYOUT=$v COT_KERNEL_SUMMARY=$O/r05_ksum_cm$v.json timeout 300 python bench.py --kernels new --steps 5 --warmup 3 --settle-seconds 3 --no-cpu-baseline --no-secondary > /dev/null 2> $O/r05_ksum_cm$v.err
done

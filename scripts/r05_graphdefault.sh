#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp; O=gpurun_out
for t in "default:COT_X=1:" "eager:COT_X=1:--eager" "failcap:COT_BENCH_FAIL_CAPTURE=1:" "coll:COT_X=1:--force-collectives" "collfail:COT_BENCH_FAIL_CAPTURE=1:--force-collectives" "recipe:COT_X=1:--recipe" "default2:COT_X=1:" "eager2:COT_X=1:--eager"; do
  name=${t%%:*}; rest=${t#*:}; envs=${rest%%:*}; flags=${rest#*:}
  ( for kv in ${envs//,/ }; do export $kv; done
    timeout 300 python bench.py $flags --steps 20 --warmup 5 --settle-seconds 6 --no-cpu-baseline --no-kernel-timing --no-secondary --no-pmc 2> $O/r05_gd_$name.err | grep '^{"metric' > $O/r05_gd_$name.json || tail -4 $O/r05_gd_$name.err )
  python -c "
import json
d=json.load(open('$O/r05_gd_$name.json')); print('$name', d['value'], d['ms_per_step'], d['final_loss'], d['host_issue_ms_per_step'], (d.get('graph') or '')[:70])"
done

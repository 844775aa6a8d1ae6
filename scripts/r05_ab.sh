#!/bin/bash
# usage: r05_ab.sh TAG "pytest files" "name:ENV=VAL[,ENV=VAL]:bench flags" ...  -- GPU parity tests, then alternating whole-step runs
mkdir -p gpurun_out; export TMPDIR=/tmp; O=gpurun_out
TAG=$1; TESTS=$2; shift 2
if [ -n "$TESTS" ]; then
  timeout 900 python -m pytest $TESTS -m gpu -q --timeout 300 -rfE -p no:cacheprovider --tb=short > $O/${TAG}_pytest.log 2>&1; echo "pytest rc=$?" >> $O/${TAG}_pytest.log
  grep -E "passed|failed|^FAILED|^ERROR|rc=|Error" $O/${TAG}_pytest.log | cut -c1-300 | tail -12
fi
for rep in 1 2; do
for t in "$@"; do
  name=${t%%:*}; rest=${t#*:}; envs=${rest%%:*}; flags=${rest#*:}
  ( for kv in ${envs//,/ }; do export $kv; done
    timeout 300 python bench.py --kernels new $flags --steps 20 --warmup 6 --settle-seconds 6 --no-cpu-baseline --no-kernel-timing --no-secondary > $O/${TAG}_step_${name}_$rep.json 2> $O/${TAG}_step_${name}_$rep.err || tail -4 $O/${TAG}_step_${name}_$rep.err )
  python -c "
import json
d=json.load(open('$O/${TAG}_step_${name}_$rep.json')); print('$name rep$rep', d['value'], d['ms_per_step'], d['final_loss'])"
done; done

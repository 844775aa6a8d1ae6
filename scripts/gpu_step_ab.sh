#!/bin/bash
# usage: gpu_step_ab.sh TAG "pytest files" "name:flags" ...   -- parity tests, then whole-step A/B runs of bench.py --kernels new
mkdir -p gpurun_out
export TMPDIR=/tmp
O=gpurun_out
TAG=$1; TESTS=$2; shift 2
if [ -n "$TESTS" ]; then
  timeout 900 python -m pytest $TESTS -m gpu -q --timeout 300 -rfE -p no:cacheprovider > $O/${TAG}_pytest.log 2>&1; echo "pytest rc=$?" >> $O/${TAG}_pytest.log
  grep -E "passed|failed|^FAILED|^ERROR|rc=" $O/${TAG}_pytest.log | cut -c1-300 | tail -10
fi
for t in "$@"; do
  name=${t%%:*}; flags=${t#*:}
  timeout 300 python bench.py --kernels new $flags --steps 20 --warmup 6 --no-cpu-baseline --no-kernel-timing > $O/${TAG}_step_$name.json 2> $O/${TAG}_step_$name.err || tail -3 $O/${TAG}_step_$name.err
  python -c "
import json
d=json.load(open('$O/${TAG}_step_$name.json')); print('$name', d['value'], d['ms_per_step'], d['final_loss'], d['config'].get('tune'))"
done

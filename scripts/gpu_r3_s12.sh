#!/bin/bash
# round 3, session 12: the default bench line (probe included), the per-block forward-only test, recipe line
mkdir -p gpurun_out; export TMPDIR=/tmp; O=gpurun_out
timeout 600 python -m pytest tests/test_fused_bn_gpu.py -m gpu -q --timeout 600 -p no:cacheprovider -k "forward_only" > $O/r3s12_pytest.log 2>&1; echo "pytest rc=$?" >> $O/r3s12_pytest.log
grep -E "passed|failed|^FAILED|^E  |rc=" $O/r3s12_pytest.log | cut -c1-600 | tail -10
( time timeout 900 python bench.py > $O/r3s12_bench_default.json 2> $O/r3s12_bench_default.err ) 2> $O/r3s12_bench_default.time
python -c "
import json
d=json.load(open('$O/r3s12_bench_default.json')); print('DEFAULT', d['value'], d['ms_per_step'], d['final_loss']); ks=d['config']['kernel_selection']; print(json.dumps(ks)[:1500]); print(json.dumps(d['roofline'])[:600]); print(json.dumps(d.get('cpu_baseline'))[:300])"
tail -3 $O/r3s12_bench_default.time
( time timeout 300 python bench.py --no-cpu-baseline > $O/r3s12_bench_cached.json 2> $O/r3s12_bench_cached.err ) 2> $O/r3s12_bench_cached.time; tail -3 $O/r3s12_bench_cached.time
python -c "
import json
d=json.load(open('$O/r3s12_bench_cached.json')); print('CACHED', d['value'], d['ms_per_step'], json.dumps(d['config']['kernel_selection'])[:200])"

#!/bin/bash
# full GPU suite + the default bench invocation (auto probe) + smoke
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
O=gpurun_out
timeout 900 python -m pytest tests -m gpu -q --timeout 300 -rfE -p no:cacheprovider > $O/r2m_pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/r2m_pytest_gpu.log
grep -E "passed|failed|^FAILED|^ERROR|rc=" $O/r2m_pytest_gpu.log | cut -c1-300 | tail -20
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $O/r2m_smoke.log 2>&1; tail -2 $O/r2m_smoke.log | cut -c1-300
/usr/bin/time -v timeout 1200 python bench.py > $O/r2m_bench_default.json 2> $O/r2m_bench_default.err; cut -c1-2500 $O/r2m_bench_default.json; grep -E "kernel set|Elapsed" $O/r2m_bench_default.err | cut -c1-1500

#!/bin/bash
# evidence session (round 6): full GPU suite, smoke, the default bench line (in-run PMC traffic, secondary configurations incl. the
# aggregation_zeropad_mix op shape), kernel trace of the default step (two streams, replayed) and of the single-stream eager step
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
O=gpurun_out
T=${1:-ev6}
T0=$(date +%s)
timeout 1500 python -m pytest tests -m gpu -q --timeout 300 -rfEx -p no:cacheprovider > $O/${T}_pytest.log 2>&1; echo "pytest rc=$? wall=$(( $(date +%s) - T0 ))s" >> $O/${T}_pytest.log
grep -E "passed|failed|^FAILED|^ERROR|^XFAIL|^XPASS|rc=" $O/${T}_pytest.log | cut -c1-300 | tail -14
timeout 180 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/${T}_smoke.log 2>&1; tail -2 $O/${T}_smoke.log | cut -c1-400
T1=$(date +%s)
timeout 900 python bench.py > $O/${T}_bench_default.json 2> $O/${T}_bench_default.err; echo "bench rc=$? wall=$(( $(date +%s) - T1 ))s"
cut -c1-400 $O/${T}_bench_default.json
if [ "$2" != "notrace" ]; then
bash scripts/gpu_trace_new.sh ${T}_trace --no-pmc > $O/${T}_trace_sh.log 2>&1; tail -2 $O/${T}_trace_sh.log | cut -c1-200
COT_WGRAD_STREAM=0 bash scripts/gpu_trace_new.sh ${T}_single --no-pmc --eager > $O/${T}_single_trace_sh.log 2>&1; tail -2 $O/${T}_single_trace_sh.log | cut -c1-200
fi
echo "session wall=$(( $(date +%s) - T0 ))s"

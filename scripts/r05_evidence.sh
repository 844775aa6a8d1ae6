#!/bin/bash
# evidence session (round 5): full GPU suite, smoke, kernel trace + stats of the default step, per-op benches through the C ABI,
# the default bench line (in-run PMC traffic, secondary configurations) and the recipe line
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
O=gpurun_out
R=$GRAFT_REPO_ROOT
T0=$(date +%s)
timeout 1200 python -m pytest tests -m gpu -q --timeout 300 -rfEx -p no:cacheprovider > $O/ev5_pytest.log 2>&1; echo "pytest rc=$? wall=$(( $(date +%s) - T0 ))s" >> $O/ev5_pytest.log
grep -E "passed|failed|^FAILED|^ERROR|^XFAIL|^XPASS|rc=" $O/ev5_pytest.log | cut -c1-300 | tail -14
timeout 180 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/ev5_smoke.log 2>&1; tail -2 $O/ev5_smoke.log | cut -c1-400
bash scripts/gpu_trace_new.sh ev5_trace --no-pmc > $O/ev5_trace_sh.log 2>&1; tail -3 $O/ev5_trace_sh.log | cut -c1-200
COT_WGRAD_STREAM=0 bash scripts/gpu_trace_new.sh ev5s --no-pmc --eager > $O/ev5s_trace_sh.log 2>&1; grep -n "agg_bwd_nchw_k3_dot2" $O/ev5s_per_shape.csv | cut -c1-140
bash scripts/r05_pmc_step.sh lds7 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS > /dev/null 2>&1
timeout 200 python scripts/bench_agg_abi.py --variants lds,dot2 --iters 20 --rounds 5 --out $O/ev5_agg_abi.json > $O/ev5_agg_abi.log 2>&1; tail -12 $O/ev5_agg_abi.log | cut -c1-200
timeout 300 python scripts/bench_conv_abi.py --iters 20 --json $O/ev5_conv_abi.json > $O/ev5_conv_abi.log 2>&1; tail -3 $O/ev5_conv_abi.log | cut -c1-200
timeout 300 python scripts/probe_cnhw.py 30 > $O/ev5_probe_cnhw.log 2>&1; grep "^sum" $O/ev5_probe_cnhw.log | cut -c1-300
T1=$(date +%s)
timeout 900 python bench.py > $O/ev5_bench_default.json 2> $O/ev5_bench_default.err; echo "bench rc=$? wall=$(( $(date +%s) - T1 ))s"
cut -c1-400 $O/ev5_bench_default.json
timeout 300 python bench.py --recipe --no-cpu-baseline --no-secondary --no-pmc > $O/ev5_bench_recipe.json 2> $O/ev5_bench_recipe.err; cut -c1-300 $O/ev5_bench_recipe.json
timeout 300 python bench.py --eager --no-cpu-baseline --no-secondary --no-pmc --no-kernel-timing > $O/ev5_bench_eager.json 2> $O/ev5_bench_eager.err; cut -c1-300 $O/ev5_bench_eager.json
echo "session wall=$(( $(date +%s) - T0 ))s"

#!/bin/bash
# evidence session (round 4): full GPU suite, smoke, kernel trace of the default step, PMC traffic of the aggregation kernels,
# per-op benches through the C ABI, the default bench line (with its secondary configurations) and the forward-only line
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
O=gpurun_out
R=$GRAFT_REPO_ROOT
T0=$(date +%s)
timeout 900 python -m pytest tests -m gpu -q --timeout 300 -rfE -p no:cacheprovider > $O/ev_pytest.log 2>&1; echo "pytest rc=$? wall=$(( $(date +%s) - T0 ))s" >> $O/ev_pytest.log
grep -E "passed|failed|^FAILED|^ERROR|rc=" $O/ev_pytest.log | cut -c1-300 | tail -12
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/ev_smoke.log 2>&1; tail -1 $O/ev_smoke.log
bash scripts/gpu_trace_new.sh ev_trace > $O/ev_trace_sh.log 2>&1; tail -3 $O/ev_trace_sh.log | cut -c1-200
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 90 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/$O/ev_pmc_$c -o pmc -- python $R/scripts/bench_agg_abi.py --shapes 0 --dtypes bf16 --variants dot2 --iters 4 --rounds 1 > $R/$O/ev_pmc_$c.log 2>&1
done
cd $R
python scripts/agg_traffic_from_pmc.py $O/ev_pmc_FETCH_SIZE $O/ev_pmc_WRITE_SIZE --out $O/ev_agg_traffic.json
python scripts/agg_traffic_from_pmc.py $O/ev_pmc_FETCH_SIZE $O/ev_pmc_WRITE_SIZE --out profiles/agg_traffic.json > /dev/null   # the bench line below reports THIS session's counters
timeout 200 python scripts/bench_agg_abi.py --variants lds,dot2 --iters 20 --rounds 5 --out $O/ev_agg_abi.json > $O/ev_agg_abi.log 2>&1; tail -20 $O/ev_agg_abi.log | cut -c1-200
timeout 300 python scripts/bench_conv_abi.py --iters 20 --json $O/ev_conv_abi.json > $O/ev_conv_abi.log 2>&1; tail -5 $O/ev_conv_abi.log | cut -c1-200
T1=$(date +%s)
timeout 900 python bench.py > $O/ev_bench_default.json 2> $O/ev_bench_default.err; echo "bench rc=$? wall=$(( $(date +%s) - T1 ))s"
cut -c1-600 $O/ev_bench_default.json
timeout 300 python bench.py --mode fwd --no-cpu-baseline > $O/ev_bench_fwd.json 2> $O/ev_bench_fwd.err || tail -5 $O/ev_bench_fwd.err; cut -c1-300 $O/ev_bench_fwd.json
timeout 300 python bench.py --mode fwd --graph --no-cpu-baseline --no-kernel-timing > $O/ev_bench_fwd_graph.json 2> $O/ev_bench_fwd_graph.err || tail -5 $O/ev_bench_fwd_graph.err; cut -c1-200 $O/ev_bench_fwd_graph.json
timeout 300 python bench.py --recipe --no-cpu-baseline > $O/ev_bench_recipe.json 2> $O/ev_bench_recipe.err; cut -c1-300 $O/ev_bench_recipe.json
echo "session wall=$(( $(date +%s) - T0 ))s"

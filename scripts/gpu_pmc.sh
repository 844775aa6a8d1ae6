#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 55 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_$c -o pmc -- python $GRAFT_REPO_ROOT/scripts/bench_agg_abi.py --shapes 0 --dtypes bf16 --variants v3d --iters 4 --rounds 1 > $GRAFT_REPO_ROOT/gpurun_out/pmc_$c.log 2>&1
done
cd $GRAFT_REPO_ROOT
find gpurun_out -name "*counter_collection.csv" | head; for f in $(find gpurun_out -name "*counter_collection.csv"); do grep -E "agg_(fwd|bwd)" $f | head -4 | cut -c1-400; done
du -sh gpurun_out

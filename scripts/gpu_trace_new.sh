#!/bin/bash
# kernel trace of the timed region of `bench.py --kernels new [extra bench arguments = $2...]` -> per-(kernel, shape) CSV in gpurun_out/$1
set -x
TAG=${1:-trace}
shift
mkdir -p gpurun_out
export TMPDIR=/tmp
O=gpurun_out
rm -rf /tmp/prof/out; mkdir -p /tmp/prof && cd /tmp/prof && COT_ROCTX=1 timeout 400 rocprofv3 --kernel-trace --marker-trace --stats --output-format csv \
    -d /tmp/prof/out -o trace -- python $GRAFT_REPO_ROOT/bench.py --kernels new --steps 5 --warmup 4 --no-cpu-baseline --no-kernel-timing --no-secondary "$@" \
    > $GRAFT_REPO_ROOT/$O/${TAG}_prof.log 2>&1
cd $GRAFT_REPO_ROOT
python scripts/trace_summary.py /tmp/prof/out --steps 5 --out $O/${TAG}_per_shape.csv; head -12 $O/${TAG}_per_shape.csv | cut -c1-180; tail -1 $O/${TAG}_per_shape.csv
for f in $(find /tmp/prof/out -name "*kernel_stats*.csv"); do cp $f $O/${TAG}_$(basename $f); done

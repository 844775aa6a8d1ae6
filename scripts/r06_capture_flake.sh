#!/bin/bash
# how often does the N > 1 (world-of-one RCCL) capture die, and with what?  usage: r06_capture_flake.sh <runs> [extra bench args]
mkdir -p gpurun_out; R=${1:-10}; shift
ok=0; bad=0
for i in $(seq $R); do
  python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-timing --no-secondary --no-pmc --settle-seconds 1 --force-collectives "$@" > /tmp/_f.json 2> /tmp/_f.err
  rc=$?
  if grep -q '^{' /tmp/_f.json; then ok=$((ok+1)); else bad=$((bad+1)); echo "--- run $i rc=$rc"; grep -v "^frame\|amdgpu.ids" /tmp/_f.err | tail -12 | cut -c1-400; fi
done
echo "runs $R ok $ok failed $bad ($*)"

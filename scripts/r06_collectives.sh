#!/bin/bash
# N > 1 step forms on ONE GPU through the world-of-one RCCL communicator (bench.py --force-collectives): whole step incl. the
# all-reduces and SGD captured / forward+backward captured + collectives eagerly / eager; against the step without a process group
mkdir -p gpurun_out
O=gpurun_out/r06_collectives.log
: > $O
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing --no-secondary --no-pmc"
run() { tag=$1; shift; "$@" > /tmp/_c.json 2> /tmp/_c.err; python - "$tag" <<'PY' >> gpurun_out/r06_collectives.log
import json,sys
try:
    d=json.loads([l for l in open('/tmp/_c.json') if l.startswith('{')][-1])
    print(f"{sys.argv[1]:34s} {d['ms_per_step']:7.3f} ms/step  {d['value']:8.1f} img/s  host issue {d['host_issue_ms_per_step']:6.2f} ms  graph: {d.get('graph','')[:150]}")
except Exception as e:
    print(sys.argv[1], 'FAILED', e, open('/tmp/_c.err').read()[-600:])
PY
}
for rep in 1 2; do
run "no group (graph)" $B
run "rccl world-of-one: full graph" $B --force-collectives
run "rccl world-of-one: deferred" $B --force-collectives --graph-collectives off
run "rccl world-of-one: eager" $B --force-collectives --eager
done
COT_BENCH_FAIL_CAPTURE=full run "fallback test (full capture fails)" $B --force-collectives
cat $O

#!/bin/bash
# the N > 1 step (world-of-one RCCL on one GPU, deferred form = the default) by bucket size and reduction dtype, against the step without a group
mkdir -p gpurun_out
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing --no-secondary --no-pmc"
run() { tag=$1; shift; "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('%-46s %.3f ms/step  host issue %.2f ms' % ('$tag', d['ms_per_step'], d['host_issue_ms_per_step']))"; }
for rep in 1 2; do
run "no group" $B
for mb in 10 25 50 100; do run "world-of-one fp32 buckets of $mb MiB" $B --force-collectives --bucket-mb $mb; done
run "world-of-one bf16 buckets of 10 MiB" $B --force-collectives --grad-dtype param
run "world-of-one bf16 buckets of 50 MiB" $B --force-collectives --grad-dtype param --bucket-mb 50
done

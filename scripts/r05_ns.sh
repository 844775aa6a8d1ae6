#!/bin/bash
# GroupNorm backward's dgamma / dbeta launch beside the weight gradients (side stream): GPU tests of the fused nodes, then the whole step vs the previous commit's library
mkdir -p gpurun_out; export TMPDIR=/tmp; O=gpurun_out
timeout 600 python -m pytest tests/test_fused_layer_gpu.py tests/test_layers_gpu.py tests/test_prepack_gpu.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -3
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1 | cut -c1-300
for rep in 1 2 3; do
  timeout 300 python bench.py --kernels new --steps 20 --warmup 6 --settle-seconds 5 --no-cpu-baseline --no-kernel-timing --no-secondary --no-pmc > $O/gnside_step_$rep.json 2> $O/gnside_step_$rep.err || tail -4 $O/gnside_step_$rep.err
  python -c "
import json
d=json.load(open('$O/gnside_step_$rep.json')); print('new rep$rep', d['value'], d['ms_per_step'], d['final_loss'])"
  timeout 300 python bench.py --kernels new --eager --steps 20 --warmup 6 --settle-seconds 5 --no-cpu-baseline --no-kernel-timing --no-secondary --no-pmc > $O/gnside_eager_$rep.json 2> $O/gnside_eager_$rep.err || tail -4 $O/gnside_eager_$rep.err
  python -c "
import json
d=json.load(open('$O/gnside_eager_$rep.json')); print('new eager rep$rep', d['value'], d['ms_per_step'], d['final_loss'])"
done | tee $O/gnside_step.log

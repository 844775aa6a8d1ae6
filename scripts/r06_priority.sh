mkdir -p gpurun_out; export TMPDIR=/tmp
python -c "import torch; print('priority range', torch.cuda.Stream.priority_range())"
run() { env $1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing --no-secondary --no-pmc $2 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('%-60s %.3f ms/step' % ('$1 $2', d['ms_per_step']))"; }
for i in 1 2; do
run "COT_MAIN_PRIORITY=0 COT_SIDE_PRIORITY=0" ""
run "COT_MAIN_PRIORITY=-1 COT_SIDE_PRIORITY=0" ""
run "COT_MAIN_PRIORITY=-1 COT_SIDE_PRIORITY=1" ""
run "COT_MAIN_PRIORITY=0 COT_SIDE_PRIORITY=0" "--eager"
run "COT_MAIN_PRIORITY=-1 COT_SIDE_PRIORITY=1" "--eager"
done

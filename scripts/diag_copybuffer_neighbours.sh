#!/bin/bash
# which kernels run right before / after the small rocclr copyBuffer launches of a step?  (kernel trace, stream order)
export TMPDIR=/tmp; mkdir -p gpurun_out; R=$GRAFT_REPO_ROOT
rm -rf /tmp/prof2; mkdir -p /tmp/prof2 && cd /tmp/prof2 && timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof2/out -o t -- \
  python $R/bench.py --kernels new --steps 2 --warmup 2 --settle-seconds 0 --no-cpu-baseline --no-kernel-timing --no-secondary > /tmp/prof2/log 2>&1
cd $R
python - <<'PY'
import csv, glob, re, collections
f = glob.glob('/tmp/prof2/out/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
names = [re.sub(r'<.*|\(.*', '', r['Kernel_Name']).replace('void ', '').replace('cot::', '')[:40] for r in rows]
prev = collections.Counter(); nxt = collections.Counter()
idx = [i for i, n in enumerate(names) if 'copyBuffer' in n]
print(len(rows), 'kernels', len(idx), 'copyBuffer')
for i in idx:
    prev[names[i-1] if i else ''] += 1
    nxt[names[i+1] if i+1 < len(names) else ''] += 1
print('before:', prev.most_common(12))
print('after :', nxt.most_common(12))
last = idx[len(idx)//2] if idx else 0
print(names[last-12:last+12])
q = collections.Counter(r['Queue_Id'] for r in rows if 'copyBuffer' in r['Kernel_Name'])
print('queues', q, collections.Counter(r['Queue_Id'] for r in rows).most_common(4))
PY

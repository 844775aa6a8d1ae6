#!/bin/bash
# SQ counters of the LDS 3x3 kernels on the CoTNet-50 layer shapes (one pass per counter group; --kernel-trace only)
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd /tmp
i=0
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_MFMA" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_WAIT_INST_LDS" "SQ_WAIT_ANY SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE SQ_LDS_DATA_FIFO_FULL"; do
  i=$((i+1))
  timeout 120 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d /tmp/pmc3/$i -o p -- python $R/scripts/bench_conv_abi.py --iters 3 --modes 1 --only "g4" > /tmp/pmc3_$i.log 2>&1
done
python - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('/tmp/pmc3/*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        if 'conv3x3g_lds' not in k: continue
        key = (k.split('(')[0][-60:], r['Grid_Size'])
        acc[key][r['Counter_Name']].append(float(r['Counter_Value']))
for key, c in sorted(acc.items()):
    print(key)
    print('   ', {n: round(sum(v)/len(v)) for n, v in sorted(c.items())})
PY

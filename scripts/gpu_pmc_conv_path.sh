#!/bin/bash
# memory-PATH counters (texture addresser / data return / L1 / LDS FIFOs) of the 1x1 kernels on ONE deep shape: where do the copies wait?
mkdir -p gpurun_out /tmp/pmc
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
cd /tmp/pmc
P1="TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_TOTAL_WAVEFRONTS_sum TA_FLAT_READ_LDS_WAVEFRONTS_sum"
P2="TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TD_TCP_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_GATE_EN1_sum TCP_TCP_LATENCY_sum"
P3="TD_TD_BUSY_sum TD_TC_STALL_sum TD_SPI_STALL_sum TD_LOAD_WAVEFRONT_sum SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_BUSY_CYCLES"
SHAPE="${1:-s3 conv1 }"
TAG=$(echo $SHAPE | tr -d ' ')
i=0
for P in "$P1" "$P2" "$P3"; do
  i=$((i+1))
  timeout 150 rocprofv3 --kernel-trace --pmc $P --output-format csv -d /tmp/pmc/p_${TAG}_$i -o p -- python $GRAFT_REPO_ROOT/scripts/bench_conv_abi.py --iters 2 --modes 1 --only "$SHAPE" > $O/pmcp_${TAG}_$i.log 2>&1
  f=$(find /tmp/pmc/p_${TAG}_$i -name "*counter_collection.csv" | head -1)
  python - "$f" "$TAG pass$i" >> $O/r3_pmc_conv_path.txt <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(collections.Counter)
for r in rows:
    k = r["Kernel_Name"][:60]
    if "lds_fwd2" not in k and "wgrad_lds2" not in k: continue
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k][r["Counter_Name"]] += 1
for k, d in agg.items():
    print(sys.argv[2], "|", k, "|", " ".join(f"{n}={v/max(cnt[k][n],1):.0f}" for n, v in sorted(d.items())))
PY
done

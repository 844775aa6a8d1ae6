#!/bin/bash
# HBM traffic of the BatchNorm-folded radix tail kernels from the PMC counters (separate --pmc passes, kernel trace only; FETCH_SIZE doubled per
# the gfx950 note of MI355X_MICROARCH.md), per CoTNet-50 stage shape at B = 80 -> gpurun_out/r06_bn_tail_traffic.log
mkdir -p gpurun_out
export TMPDIR=/tmp
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c; rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_$c -o pmc -- python $GRAFT_REPO_ROOT/scripts/bench_bn_tail.py > /tmp/pmc_$c.log 2>&1; tail -3 /tmp/pmc_$c.log | cut -c1-300
done
python - <<'PY' 2>&1 | tee $GRAFT_REPO_ROOT/gpurun_out/r06_bn_tail_traffic.log
import csv, glob, os
names = {"bn_stats_sums": 1, "radix_gap_t_bn_kernel": 2, "radix_mix_logits_bn_kernel": 3, "radix_mix_bwd_reduce_bn_kernel": 3, "radix_mix_bwd_apply_bn_kernel": 4}  # tensor passes
shapes = {5120: (80, 64, 3136), 10240: (80, 128, 784), 20480: (80, 256, 196), 40960: (80, 512, 49)}
def per(root, counter):
    acc = {}
    for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] != counter:
                continue
            k = [n for n in names if n in row["Kernel_Name"]]
            if k:
                acc.setdefault((k[0], row["Grid_Size"], "Li7ELi8E" in row["Kernel_Name"] or ", 7, 8>" in row["Kernel_Name"]), []).append(float(row["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}
f, w = per("/tmp/pmc_FETCH_SIZE", "FETCH_SIZE"), per("/tmp/pmc_WRITE_SIZE", "WRITE_SIZE")
print("kernel, grid: HBM bytes per launch from the counters (2 x FETCH_SIZE + WRITE_SIZE, KB units) against tensor passes x bf16 tensor size")
for (k, grid, p7) in sorted(f, key=lambda kg: (kg[0], kg[2], -int(kg[1]))):
    if (k, grid, p7) not in w:
        continue
    b = 2 * f[(k, grid, p7)] * 1024 + w[(k, grid, p7)] * 1024
    g = int(grid)
    planes = [p for p in shapes if (p == 40960) == p7 and g in (p * 64, ((p + 31) // 32) * 256)]
    note = ""
    if planes and k != "bn_stats_sums":
        N, C, HW = shapes[planes[0]]
        alg = names[k] * N * C * HW * 2
        note = f" = {b / alg:.2f} x the algorithmic {alg / 1e6:.1f} MB (N{N} C{C} HW{HW}, {names[k]} passes)"
    print(f"{k:34s} grid {grid:>9s}: {b / 1e6:8.2f} MB{note}")
PY

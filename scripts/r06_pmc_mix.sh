#!/bin/bash
# HBM traffic of the mix tile kernels from the PMC counters (separate --pmc passes, kernel trace only; FETCH_SIZE doubled per the gfx950
# note of MI355X_MICROARCH.md) at the op shape (B = 64, C = 256, 20 x 20, wC = 32) -> gpurun_out/r06_aggmix_traffic.log
mkdir -p gpurun_out
export TMPDIR=/tmp
cd /tmp
for dt in bf16 fp32; do
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c; rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_$c -o pmc -- python $GRAFT_REPO_ROOT/scripts/bench_aggmix_abi.py --dtypes $dt --iters 4 --rounds 1 > /tmp/pmc_$c.log 2>&1
done
python - "$dt" <<'PY'
import csv, glob, os, sys
dt = sys.argv[1]
esz = 2 if dt == "bf16" else 4
alg = esz * 64 * 400 * (256 + 34 * 32 + 512)
def per(root, counter):
    acc = {}
    for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] == counter and "aggmix" in row["Kernel_Name"]:
                k = [n for n in ("aggmix_fwd_tile", "aggmix_bwd_input_tile", "aggmix_bwd_weight_tile") if n in row["Kernel_Name"]]
                if k:
                    acc.setdefault(k[0], []).append(float(row["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}
f, w = per("/tmp/pmc_FETCH_SIZE", "FETCH_SIZE"), per("/tmp/pmc_WRITE_SIZE", "WRITE_SIZE")
for k in sorted(f):
    if k in w:
        b = 2 * f[k] * 1024 + w[k] * 1024
        print(f"{dt} {k:24s} FETCH_SIZE {f[k]:9.1f} KB  WRITE_SIZE {w[k]:9.1f} KB  -> {b / 1e6:7.2f} MB per launch = {b / alg:.3f} x the algorithmic {alg / 1e6:.1f} MB")
PY
done 2>&1 | tee $GRAFT_REPO_ROOT/gpurun_out/r06_aggmix_traffic.log

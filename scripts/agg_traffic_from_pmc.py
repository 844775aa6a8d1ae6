#!/usr/bin/env python
"""HBM-side traffic per launch of the aggregation kernels from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) over
scripts/bench_agg_abi.py, corrected as /opt/skills/guides/MI355X_MICROARCH.md (HBM section) prescribes for gfx950:
bytes = 2 * FETCH_SIZE[KB] * 1024 + WRITE_SIZE[KB] * 1024.  Writes the {"kernel|shape|dtype": bytes} table bench.py reports as
roofline.traffic.

    python scripts/agg_traffic_from_pmc.py <dir with FETCH_SIZE pass> <dir with WRITE_SIZE pass> --shape N80xC64x56x56 \
        --dtype bfloat16 --out profiles/agg_traffic.json
"""
import argparse
import csv
import glob
import json
import os


def per_kernel(root, counter):
    acc = {}
    for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] != counter:
                continue
            name = row["Kernel_Name"]
            for k in ("agg_fwd_nchw_k3_lds", "agg_bwd_nchw_k3_lds", "agg_bwd_nchw_k3_dot2"):
                if k in name:
                    acc.setdefault(k, []).append(float(row["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}, {k: len(v) for k, v in acc.items()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("fetch_dir")
    ap.add_argument("write_dir")
    ap.add_argument("--shape", default="N80xC64x56x56")
    ap.add_argument("--dtype", default="bfloat16")
    ap.add_argument("--out", default="")
    a = ap.parse_args()
    fetch, nf = per_kernel(a.fetch_dir, "FETCH_SIZE")
    write, nw = per_kernel(a.write_dir, "WRITE_SIZE")
    table = {}
    for k in sorted(fetch):
        if k in write:
            table[f"{k}|{a.shape}|{a.dtype}"] = int(round(2 * fetch[k] * 1024 + write[k] * 1024))
            print(f"{k}: FETCH_SIZE {fetch[k]:.1f} KB x{nf[k]} launches, WRITE_SIZE {write[k]:.1f} KB x{nw[k]} -> "
                  f"{table[f'{k}|{a.shape}|{a.dtype}'] / 1e6:.2f} MB per launch")
    if a.out:
        old = json.load(open(a.out)) if os.path.exists(a.out) else {}
        old.update(table)
        json.dump(old, open(a.out, "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()

#!/usr/bin/env python
"""Condense a `rocprofv3 --kernel-trace` CSV into one row per (kernel, grid, workgroup) = per kernel AND shape:
launches, mean / min / max duration, total time and share.  rocprofv3's own --stats table lumps all shapes of a
template into one row (verdict r1, weak #5); the roofline of a kernel needs its per-shape time.

    python scripts/trace_summary.py <dir-or-kernel_trace.csv> [--steps K] [--out profiles/rNN_....csv]
"""
import argparse
import csv
import glob
import os
import re
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    name = name.replace("cot::", "")
    m = re.match(r"([\w:]+)(<.*>)?", name)
    base = m.group(1) if m else name
    targs = (m.group(2) or "") if m else ""
    targs = re.sub(r"__hip_bfloat16|hip_bfloat16|__bf16", "bf16", targs)
    if len(targs) > 60:
        targs = targs[:57] + "...>"
    return base + targs


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("path")
    ap.add_argument("--steps", type=int, default=1, help="steps covered by the trace (per-step columns)")
    ap.add_argument("--out", default=None)
    ap.add_argument("--top", type=int, default=0)
    ap.add_argument("--all", action="store_true", help="every launch of the trace (default: the largest burst = the timed region)")
    a = ap.parse_args()
    files = [a.path] if os.path.isfile(a.path) else glob.glob(os.path.join(a.path, "**", "*kernel_trace.csv"), recursive=True)
    if not files:
        sys.exit(f"no *kernel_trace.csv under {a.path}")
    rows = {}
    recs = []
    for f in files:
        with open(f, newline="") as fh:
            recs.extend(csv.DictReader(fh))
    if not a.all and recs:
        # bench.py's COT_ROCTX window opens the collection for the timed region only, but the profiler also records what ran before the
        # first pause (model construction: hundreds of small copyBuffer / fill launches).  Keep the burst with the most kernels = the timed
        # replays (bursts = runs of launches with less than 2 ms of silence between them)
        recs.sort(key=lambda r: int(r["Start_Timestamp"]))
        bursts, cur, last_end = [], [], None
        for r in recs:
            if last_end is not None and int(r["Start_Timestamp"]) - last_end > 2_000_000:
                bursts.append(cur)
                cur = []
            cur.append(r)
            last_end = max(last_end or 0, int(r["End_Timestamp"]))
        bursts.append(cur)
        keep = max(bursts, key=len)
        print(f"{len(recs)} launches in {len(bursts)} bursts {[len(b) for b in bursts]}: keeping the largest ({len(keep)})", file=sys.stderr)
        recs = keep
    for r in recs:
        dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3  # us
        grid = "x".join(r.get(k, "1") for k in ("Grid_Size_X", "Grid_Size_Y", "Grid_Size_Z"))
        wg = "x".join(r.get(k, "1") for k in ("Workgroup_Size_X", "Workgroup_Size_Y", "Workgroup_Size_Z"))
        key = (short(r["Kernel_Name"]), grid, wg)
        e = rows.setdefault(key, {"n": 0, "tot": 0.0, "min": 1e30, "max": 0.0, "vgpr": r.get("VGPR_Count", ""),
                                  "agpr": r.get("Accum_VGPR_Count", ""), "lds": r.get("LDS_Block_Size", ""),
                                  "scratch": r.get("Scratch_Size", "")})
        e["n"] += 1
        e["tot"] += dur
        e["min"] = min(e["min"], dur)
        e["max"] = max(e["max"], dur)
    total = sum(e["tot"] for e in rows.values())
    out = open(a.out, "w", newline="") if a.out else sys.stdout
    w = csv.writer(out)
    w.writerow(["kernel", "grid(threads)", "workgroup", "launches_per_step", "avg_us", "min_us", "max_us", "ms_per_step",
                "share_pct", "vgpr", "agpr", "lds", "scratch"])
    items = sorted(rows.items(), key=lambda kv: -kv[1]["tot"])
    if a.top:
        items = items[:a.top]
    for (name, grid, wg), e in items:
        w.writerow([name, grid, wg, round(e["n"] / a.steps, 2), round(e["tot"] / e["n"], 2), round(e["min"], 2),
                    round(e["max"], 2), round(e["tot"] / a.steps / 1e3, 4), round(100 * e["tot"] / total, 2), e["vgpr"],
                    e["agpr"], e["lds"], e["scratch"]])
    w.writerow(["TOTAL", "", "", round(sum(e["n"] for e in rows.values()) / a.steps, 1), "", "", "",
                round(total / a.steps / 1e3, 3), 100.0, "", "", "", ""])
    if a.out:
        out.close()


if __name__ == "__main__":
    main()

#!/usr/bin/env python
"""For every launch of a kernel matching PATTERN in a `rocprofv3 --kernel-trace` CSV: the kernels launched right before / after it (global
start order) and its queue -- where do stray runtime kernels (copyBuffer, fillBuffer) in a replayed step come from?

    python scripts/trace_neighbours.py <dir-or-kernel_trace.csv> PATTERN [--top 30]
"""
import argparse
import csv
import glob
import os
import re
import sys


def short(n):
    n = re.sub(r"^void ", "", n).replace("cot::", "")
    m = re.match(r"(_ZN3cot\d*)?([\w:]+)", n)
    return n.split("(")[0][:60]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("path")
    ap.add_argument("pattern")
    ap.add_argument("--top", type=int, default=30)
    ap.add_argument("--all", action="store_true", help="every event of the trace (default: the burst with the most kernels)")
    a = ap.parse_args()
    files = [a.path] if os.path.isfile(a.path) else glob.glob(os.path.join(a.path, "**", "*kernel_trace.csv"), recursive=True)
    ev = []
    for f in files:
        for r in csv.DictReader(open(f, newline="")):
            ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "0"), r["Kernel_Name"], r.get("Grid_Size", r.get("Grid_Size_X", "")), r.get("Stream_Id", "")))
    ev.sort()
    # the burst with the most kernels = the timed replays
    bursts, cur = [], [ev[0]]
    for x in ev[1:]:
        if x[0] - max(e[1] for e in cur[-8:]) > 2_000_000:
            bursts.append(cur)
            cur = []
        cur.append(x)
    bursts.append(cur)
    print(f'{len(ev)} kernels in {len(bursts)} bursts: {[len(b) for b in bursts]}')
    ev = ev if a.all else max(bursts, key=len)
    hist, qs, total = {}, {}, 0
    for i, e in enumerate(ev):
        if a.pattern in e[3]:
            total += 1
            prev = next((ev[j] for j in range(i - 1, -1, -1) if a.pattern not in ev[j][3]), None)
            nxt = next((ev[j] for j in range(i + 1, len(ev)) if a.pattern not in ev[j][3]), None)
            k = (short(prev[3]) if prev else "-", short(nxt[3]) if nxt else "-")
            hist[k] = hist.get(k, 0) + 1
            qs[(e[2], e[5], e[4])] = qs.get((e[2], e[5], e[4]), 0) + 1
    print(f"{total} launches matching {a.pattern!r} among {len(ev)} kernels; by (queue, stream, grid):")
    for k, v in sorted(qs.items(), key=lambda kv: -kv[1])[:12]:
        print(f"   {v:6d}  queue {k[0]} stream {k[1]} grid {k[2]}")
    print("by (kernel before -> kernel after):")
    for k, v in sorted(hist.items(), key=lambda kv: -kv[1])[:a.top]:
        print(f"   {v:6d}  {k[0]}  ->  {k[1]}")


if __name__ == "__main__":
    main()

#!/usr/bin/env python
"""Timeline view of a `rocprofv3 --kernel-trace` CSV: per hardware queue the busy time, the union of busy intervals over all queues
(= time the GPU runs at least one kernel), idle gaps on the union, and how much of the wall time two queues overlap.

    python scripts/trace_timeline.py <dir-or-kernel_trace.csv> [--steps K] [--skip-first-ms X]
"""
import argparse
import csv
import glob
import os
import sys


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("path")
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--gaps", type=int, default=0, help="list the N largest idle gaps with their neighbours")
    ap.add_argument("--verbose", action="store_true")
    a = ap.parse_args()
    files = [a.path] if os.path.isfile(a.path) else glob.glob(os.path.join(a.path, "**", "*kernel_trace.csv"), recursive=True)
    ev = []
    for f in files:
        for r in csv.DictReader(open(f, newline="")):
            ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "0"), r["Kernel_Name"]))
    ev.sort()
    if not ev:
        sys.exit("no events")
    # the trace may hold several bursts (settling / capture / timed region): keep the burst with the most kernels
    bursts, cur = [], [ev[0]]
    for x in ev[1:]:
        if x[0] - max(e[1] for e in cur[-8:]) > 2_000_000:  # > 2 ms of silence
            bursts.append(cur)
            cur = []
        cur.append(x)
    bursts.append(cur)
    print(f"{len(ev)} kernels in {len(bursts)} bursts: {[len(b) for b in bursts]}")
    ev = max(bursts, key=len)
    t0, t1 = ev[0][0], max(e[1] for e in ev)
    wall = (t1 - t0) / 1e6
    queues = {}
    for s, e, q, n in ev:
        queues.setdefault(q, []).append((s, e))
    print(f"{len(ev)} kernels over {wall:.3f} ms ({wall / a.steps:.3f} ms per step for {a.steps} steps)")
    for q, iv in sorted(queues.items(), key=lambda kv: -len(kv[1])):
        busy = sum(e - s for s, e in iv) / 1e6
        print(f"  queue {q}: {len(iv)} kernels, busy {busy:.3f} ms ({busy / a.steps:.3f} per step)")
    # union of busy intervals
    merged = []
    for s, e, _, _ in ev:
        if merged and s <= merged[-1][1]:
            merged[-1][1] = max(merged[-1][1], e)
        else:
            merged.append([s, e])
    union = sum(e - s for s, e in merged) / 1e6
    gaps = [(merged[i + 1][0] - merged[i][1]) / 1e3 for i in range(len(merged) - 1)]
    print(f"  union busy {union:.3f} ms ({union / a.steps:.3f} per step); idle {wall - union:.3f} ms in {len(gaps)} gaps: "
          f"{sum(1 for g in gaps if g < 2)} under 2 us, {sum(1 for g in gaps if 2 <= g < 10)} of 2-10 us, {sum(1 for g in gaps if g >= 10)} over 10 us "
          f"(largest {max(gaps) if gaps else 0:.1f} us)")
    # the largest idle gaps of the union: which kernel ended last before the gap, which started after it (fork / join points, graph launch)
    if a.gaps:
        ends = sorted(ev, key=lambda x: x[1])
        import bisect
        end_ts = [x[1] for x in ends]
        big = sorted(((merged[i + 1][0] - merged[i][1], merged[i][1], merged[i + 1][0]) for i in range(len(merged) - 1)), reverse=True)[:a.gaps]
        starts = {x[0]: x for x in ev}
        short = lambda n: n.split("(")[0].replace("void cot::", "").replace("cot::", "")[:46]  # noqa: E731
        hist = {}
        for g, te, ts in sorted(big, key=lambda x: x[1]):
            before = ends[bisect.bisect_right(end_ts, te) - 1]
            after = starts[ts]
            hist[(short(before[3]), short(after[3]))] = hist.get((short(before[3]), short(after[3])), 0) + g / 1e3
            if a.verbose:
                print(f"    gap {g / 1e3:6.1f} us at +{(te - t0) / 1e6:8.3f} ms: after {short(before[3])} [q{before[2]}] -> before {short(after[3])} [q{after[2]}]")
        print(f"  the {len(big)} largest gaps by (kernel before, kernel after), us summed:")
        for k, v in sorted(hist.items(), key=lambda kv: -kv[1])[:25]:
            print(f"    {v:8.1f}  {k[0]} -> {k[1]}")
    # per-queue gaps on the busiest queue: dependent-launch boundaries
    q0 = max(queues.items(), key=lambda kv: len(kv[1]))[1]
    q0.sort()
    g0 = [(q0[i + 1][0] - q0[i][1]) / 1e3 for i in range(len(q0) - 1)]
    pos = [g for g in g0 if g > 0]
    print(f"  busiest queue: gaps between consecutive kernels: median {sorted(pos)[len(pos) // 2] if pos else 0:.2f} us, sum {sum(pos) / 1e3:.3f} ms "
          f"({sum(pos) / 1e3 / a.steps:.3f} per step)")


if __name__ == "__main__":
    main()

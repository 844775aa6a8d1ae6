#!/usr/bin/env python
"""Copy what scripts/gpu_evidence_session.sh left in gpurun_out/ev_* into profiles/<round>_* and print the figures DESIGN.md
section 7.0 quotes (JSON on stdout).

    python scripts/collect_evidence.py r04 > /tmp/values.json"""
import collections
import csv
import json
import os
import re
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G, P = os.path.join(ROOT, "gpurun_out"), os.path.join(ROOT, "profiles")
rnd = sys.argv[1] if len(sys.argv) > 1 else "r04"


def line(name):
    txt = open(os.path.join(G, name)).read().strip().splitlines()
    return json.loads(next(ln for ln in reversed(txt) if ln.startswith("{")))


copies = {"ev_pytest.log": "pytest_gpu.log", "ev_smoke.log": "smoke.log", "ev_bench_default.json": "bench_default.json",
          "ev_bench_fwd.json": "bench_fwd.json", "ev_bench_fwd_graph.json": "bench_fwd_graph.json", "ev_bench_recipe.json": "bench_recipe.json", "ev_agg_abi.log": "agg_abi.log",
          "ev_conv_abi.log": "conv_abi.log", "ev_trace_per_shape.csv": "rocprofv3_kernel_trace_new_per_shape.csv",
          "ev_trace_trace_kernel_stats.csv": "rocprofv3_kernel_stats_new.csv", "ev_agg_traffic.json": "agg_traffic_session.json"}
for src, dst in copies.items():
    if os.path.exists(os.path.join(G, src)):
        shutil.copy(os.path.join(G, src), os.path.join(P, f"{rnd}_{dst}"))
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    src = os.path.join(G, f"ev_pmc_{c}", "pmc_counter_collection.csv")
    if os.path.exists(src):
        os.makedirs(os.path.join(P, f"{rnd}_pmc"), exist_ok=True)
        # (the counter CSV of the aggregation micro-benchmark only: a few hundred rows)
        shutil.copy(src, os.path.join(P, f"{rnd}_pmc", f"agg_dot2_{c}_counter_collection.csv"))

d = line("ev_bench_default.json")
r = d["roofline"]
sec = d.get("secondary", {})
v = {"HEAD_IMGS": round(d["value"]), "HEAD_MS": f"{d['ms_per_step']:.2f}", "FRAC": f"{r['frac']:.3f}", "BWD_US": f"{r['avg_us']:.1f}"}
for key, tag in (("cotnext101_2x48d_b64_224", "C4"), ("se_cotnetd_152_L_b64_320", "C5"), ("cotnet50_b80_224_fp32", "FP32")):
    e = sec.get(key, {})
    v[f"{tag}_IMGS"] = round(e["value"]) if "value" in e else "failed"
    v[f"{tag}_MS"] = f"{e['ms_per_step']:.1f}" if "ms_per_step" in e else "-"
for name, tag in (("ev_bench_fwd.json", "FWD"), ("ev_bench_fwd_graph.json", "FWDG"), ("ev_bench_recipe.json", "RECIPE")):
    if not os.path.exists(os.path.join(G, name)):
        continue
    e = line(name)
    v[f"{tag}_IMGS"], v[f"{tag}_MS"] = round(e["value"]), f"{e['ms_per_step']:.2f}"
m = re.search(r"(\d+) passed", open(os.path.join(G, "ev_pytest.log")).read())
v["NTESTS"] = m.group(1) if m else "?"
# kernel families of the traced step
fam = collections.defaultdict(float)
rules = [("bn_", "BatchNorm"), ("conv1x1_lds_fwd", "1×1 forward / data gradient"), ("conv1x1_fwd_mfma", "1×1 forward / data gradient"),
         ("conv1x1_wgrad_lds", "1×1 and 3×3 weight gradients"), ("conv3x3g_wgrad", "1×1 and 3×3 weight gradients"), ("wgrad_reduce", "weight-gradient reduces"),
         ("conv1x1_wgrad_reduce", "weight-gradient reduces"), ("conv3x3g_lds", "3×3 forward / data gradient"), ("conv3x3g_repack", "3×3 weight repack"),
         ("gn9", "GroupNorm"), ("agg_", "aggregation"), ("radix_", "radix tail"), ("tiny::", "`se` branch convolutions"),
         ("stem7x7", "stem"), ("pool3x3", "poolings + sub-sampling"), ("subsample2", "poolings + sub-sampling"), ("avgpool", "poolings + sub-sampling"),
         ("sgd_flat", "SGD"), ("copyBuffer", None), ("fillBuffer", None)]
tot = 0.0
step_ms = None
for row in csv.DictReader(open(os.path.join(G, "ev_trace_per_shape.csv"))):
    k = re.sub(r"^_ZN3cot\d+", "", row["kernel"])
    if k.startswith("TOTAL"):
        continue
    ms = float(row["ms_per_step"])
    for pat, name in rules:
        if pat in k:
            break
    else:
        name = "everything torch still launches (loss, casts)"
    if name is None:
        continue  # (model set-up copies inside the profiled window, not part of a step)
    fam[name] += ms
    tot += ms
v["KTOT"] = f"{tot:.1f}"
v["FAMILIES"] = ", ".join(f"{k} {x:.2f}" for k, x in sorted(fam.items(), key=lambda kv: -kv[1]))
try:
    t = [json.loads(ln) for ln in open(os.path.join(G, "ev_trace_prof.log")) if ln.startswith("{")]
    v["TRACE_MS"] = f"{t[-1]['ms_per_step']:.1f} ms under the profiler"
except Exception:  # noqa: BLE001
    v["TRACE_MS"] = "?"
json.dump(v, sys.stdout, indent=1, ensure_ascii=False)

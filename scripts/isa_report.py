#!/usr/bin/env python3
"""Static ISA report for the library's kernels (no GPU needed: hipcc cross-compiles for gfx950).

For every kernel of the given sources it prints the compiler's resource remarks (VGPRs / AGPRs / SGPRs, scratch bytes,
occupancy) and, from the assembly, the memory-instruction mix and the histogram of `s_waitcnt vmcnt(N)` operands.  The
last is the quick check that a register ring really keeps loads in flight: a steady state that multiplies stage k while
stages k+1.. are on their way waits with vmcnt(#loads of the younger stages), not vmcnt(0)  (DESIGN.md 4.7).

    python scripts/isa_report.py conv1x1 conv3x3g group_norm9 > profiles/rNN_isa_report.txt
    python scripts/isa_report.py --only k3_lds agg_nchw        (only kernels whose name contains the substring)
"""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "cotnet_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-gpu-rdc"]


def demangle(names):
    # binutils' c++filt does not know the mangling of __bf16 ("DF16b"): hand it over as a vendor type of the same name
    text = "\n".join(n.replace("DF16b", "u6__bf16") for n in names)
    out = subprocess.run(["c++filt"], input=text, capture_output=True, text=True).stdout.splitlines()
    return [re.sub(r"\(.*$", "", re.sub(r"^void ", "", d)) for d in out]


def resources(src, tmp):
    r = subprocess.run([HIPCC, *FLAGS, "-fPIC", "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o",
                        os.path.join(tmp, "o.o")], capture_output=True, text=True)
    res = {}
    for blk in r.stderr.split("Function Name: ")[1:]:
        name = blk.split()[0]
        def g(key):
            m = re.search(key + r": (\d+)", blk)
            return int(m.group(1)) if m else -1
        res[name] = (g("VGPRs"), g("AGPRs"), g("TotalSGPRs"), g(r"ScratchSize \[bytes/lane\]"),
                     g(r"Occupancy \[waves/SIMD\]"))
    return res


def kernels(src, tmp):
    asm = os.path.join(tmp, "k.s")
    subprocess.run([HIPCC, *FLAGS, "-S", "--cuda-device-only", "-o", asm, src], check=True, capture_output=True)
    cur, body = None, collections.OrderedDict()
    for line in open(asm):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            cur = m.group(1)
            body[cur] = []
        elif line.startswith(".Lfunc_end"):
            cur = None
        elif cur:
            body[cur].append(line)
    return body


def main():
    args, only = sys.argv[1:], None
    if args[:1] == ["--only"]:
        only, args = args[1], args[2:]
    for stem in args or ["conv1x1", "conv3x3g", "group_norm9", "stem7x7"]:
        src = os.path.join(CSRC, stem + ".hip")
        with tempfile.TemporaryDirectory() as tmp:
            res, body = resources(src, tmp), kernels(src, tmp)
        names = list(body)
        print(f"==== {stem}.hip")
        for name, dem in zip(names, demangle(names)):
            if only and only not in dem:
                continue
            lines = body[name]
            mem = collections.Counter(m.group(1) for ln in lines
                                      for m in [re.match(r"\s+((?:global|ds|buffer|scratch)_\w+)", ln)] if m)
            waits = collections.Counter(int(m.group(1)) for ln in lines
                                        for m in [re.search(r"s_waitcnt.*vmcnt\((\d+)\)", ln)] if m)
            mfma = sum(1 for ln in lines if "v_mfma" in ln)
            v, a, s, scr, occ = res.get(name, (-1,) * 5)
            print(f"-- {dem}")
            print(f"   VGPR {v} AGPR {a} SGPR {s} scratch {scr} B/lane, occupancy {occ} waves/SIMD, {len(lines)} lines, "
                  f"{mfma} v_mfma")
            print("   memory ops: " + ", ".join(f"{k} x{n}" for k, n in sorted(mem.items())))
            print("   vmcnt waits: " + ", ".join(f"({k}) x{n}" for k, n in sorted(waits.items(), reverse=True)))


if __name__ == "__main__":
    main()

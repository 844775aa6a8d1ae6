"""Every `path:line[-line]` citation of a reference file in the sources, headers and documents names an existing file of the
reference checkout and a line range inside it (run where /root/reference exists: the build container; skipped elsewhere).
Test infrastructure only: it reads the reference's line counts, nothing else."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
PAT = re.compile(r"([A-Za-z_][\w/\.]*\.(?:py|yaml|sh)):(\d+)(?:-:?(\d+))?")
OURS = ("tests/", "scripts/", "cotnet_amd/", "oracle/", "profiles/", "bench", "__graft")
NOT_OURS = ("SURVEY", "VERDICT", "ADVICE", "PAPERS", "SNIPPETS", "BASELINE")


@pytest.mark.skipif(not os.path.isdir(REF), reason="no reference checkout here")
def test_reference_citations_resolve():
    by_name = {}
    for d, _, fs in os.walk(REF):
        if ".git" in d:
            continue
        for f in fs:
            by_name.setdefault(f, []).append(os.path.join(d, f))
    lines = {}

    def nlines(p):
        if p not in lines:
            with open(p, errors="replace") as fh:
                lines[p] = sum(1 for _ in fh)
        return lines[p]

    tracked = subprocess.run(["git", "ls-files", "*.py", "*.md", "*.h", "*.hip", "*.c", "*.sh"], cwd=ROOT, capture_output=True,
                             text=True).stdout.split()
    if not tracked:
        pytest.skip("not a git checkout")
    checked, bad = 0, []
    for s in tracked:
        if s.startswith(NOT_OURS):
            continue
        with open(os.path.join(ROOT, s), errors="replace") as fh:
            txt = fh.read()
        for m in PAT.finditer(txt):
            path, last = m.group(1), int(m.group(3) or m.group(2))
            base = os.path.basename(path)
            if os.path.exists(os.path.join(ROOT, path)) or path.startswith(OURS):
                continue
            cands = [p for p in by_name.get(base, []) if p.endswith("/" + path)] or ([] if "/" in path else by_name.get(base, []))
            if not cands:
                continue  # (our own files cited by base name, abbreviations like `mix.py`)
            checked += 1
            if not any(last <= nlines(c) for c in cands):
                bad.append((s, m.group(0)))
    assert checked > 200 and not bad, (checked, bad[:10])

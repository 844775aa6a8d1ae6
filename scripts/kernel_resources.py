"""Registers / scratch / occupancy of the kernels of one source file, from hipcc's own remarks.

    python scripts/kernel_resources.py agg_mix.hip [substring of the demangled kernel name]
"""
import re
import subprocess
import sys

src = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 else ''
out = subprocess.run(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-fno-gpu-rdc', '-c', src,
                      '-o', '/tmp/_kernel_resources.o', '-Rpass-analysis=kernel-resource-usage'],
                     capture_output=True, text=True, cwd='/root/repo/cotnet_amd/csrc').stderr
cur, rows = None, {}
for line in out.splitlines():
    m = re.search(r'remark:\s+([A-Za-z ]+(?: \[[^\]]*\])?): (\S+)', line)
    if not m:
        continue
    k, v = m.groups()
    if k == 'Function Name':
        cur = subprocess.run(['c++filt', v], capture_output=True, text=True).stdout.strip()
        rows[cur] = {}
    elif cur:
        rows[cur][k] = v
for k, v in rows.items():
    if pat in k:
        print(f"{k[:100]:100s} vgpr {v.get('VGPRs', '?'):>4} agpr {v.get('AGPRs', '?'):>3} sgpr {v.get('TotalSGPRs', '?'):>3} "
              f"scratch {v.get('ScratchSize [bytes/lane]', '?'):>4} waves/SIMD {v.get('Occupancy [waves/SIMD]', '?')}")

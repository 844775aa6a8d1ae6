"""Instruction mix per kernel of one .hip source (device assembly): python scripts/isa_mix.py agg_mix.hip tile IDF16bLi4"""
import collections
import re
import subprocess
import sys

src = sys.argv[1]
pats = sys.argv[2:]
asm = '/tmp/_isa_mix.s'
subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-fno-gpu-rdc', '-S',
                       '--cuda-device-only', src, '-o', asm], cwd='/root/repo/cotnet_amd/csrc', stderr=subprocess.DEVNULL)
s = open(asm).read()
parts = re.split(r'\n(_Z\w+):[^\n]*\n', s)
for i in range(1, len(parts), 2):
    name = parts[i]
    if not all(p in name for p in pats):
        continue
    body = parts[i + 1].split('.Lfunc_end')[0]
    ops = re.findall(r'^\s+([a-z_0-9]+)', body, re.M)
    c = collections.Counter()
    for o in ops:
        if o.startswith(('global_', 'ds_', 'scratch_', 'buffer_', 'v_mfma', 'v_dot', 's_barrier', 's_waitcnt')):
            c[o] += 1
        elif o.startswith('v_'):
            c['VALU'] += 1
            if o.startswith(('v_fma', 'v_fmac', 'v_pk_fma', 'v_mac')):
                c['  fma'] += 1
        elif o.startswith('s_'):
            c['SALU'] += 1
    print(name)
    print('   ', ', '.join(f'{k} {v}' for k, v in sorted(c.items())))

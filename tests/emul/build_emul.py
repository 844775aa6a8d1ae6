"""Compile cotnet_amd/csrc/*.hip for the HOST with the shim in tests/emul/hip (test infrastructure only).

One object per source (built in parallel, rebuilt only when the source or a header changed), then one link."""
import glob
import os
import subprocess
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
OUT = os.path.join(HERE, "libcotnet_emul.so")
OBJ = os.path.join(HERE, "build")
CXX = "/opt/rocm/lib/llvm/bin/clang++"


def build():
    srcs = sorted(glob.glob(os.path.join(ROOT, "cotnet_amd", "csrc", "*.hip")))
    hdrs = glob.glob(os.path.join(ROOT, "cotnet_amd", "csrc", "*.h")) + \
        [os.path.join(ROOT, "include", "cotnet_amd.h"), os.path.join(HERE, "hip", "hip_runtime.h"), os.path.abspath(__file__)]
    if not os.path.exists(CXX):
        raise FileNotFoundError(CXX)
    os.makedirs(OBJ, exist_ok=True)
    hdr_time = max(os.path.getmtime(h) for h in hdrs)
    objs, todo = [], []
    for s in srcs:
        o = os.path.join(OBJ, os.path.basename(s)[:-4] + ".o")
        objs.append(o)
        if not os.path.exists(o) or os.path.getmtime(o) < max(os.path.getmtime(s), hdr_time):
            todo.append((s, o))
    stale = [o for o in glob.glob(os.path.join(OBJ, "*.o")) if o not in objs]
    for o in stale:
        os.remove(o)

    def cc(so):
        s, o = so
        subprocess.check_call([CXX, "-x", "c++", "-std=c++17", "-O1", "-fPIC", "-fopenmp", "-w", "-I", HERE, "-c", s, "-o", o])
    if todo:
        with ThreadPoolExecutor(max_workers=min(8, len(todo))) as ex:
            list(ex.map(cc, todo))
    if todo or stale or not os.path.exists(OUT) or any(os.path.getmtime(OUT) < os.path.getmtime(o) for o in objs):
        subprocess.check_call([CXX, "-shared", "-fPIC", "-fopenmp", "-o", OUT] + objs)
    return OUT


if __name__ == "__main__":
    print(build())

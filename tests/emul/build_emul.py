"""Compile cotnet_amd/csrc/*.hip for the HOST with the shim in tests/emul/hip (test infrastructure only)."""
import glob
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
OUT = os.path.join(HERE, "libcotnet_emul.so")
CXX = "/opt/rocm/lib/llvm/bin/clang++"


def build():
    srcs = sorted(glob.glob(os.path.join(ROOT, "cotnet_amd", "csrc", "*.hip")))
    deps = srcs + glob.glob(os.path.join(ROOT, "cotnet_amd", "csrc", "*.h")) + \
        [os.path.join(ROOT, "include", "cotnet_amd.h"), os.path.join(HERE, "hip", "hip_runtime.h")]
    if os.path.exists(OUT) and all(os.path.getmtime(OUT) >= os.path.getmtime(d) for d in deps):
        return OUT
    if not os.path.exists(CXX):
        raise FileNotFoundError(CXX)
    cmd = [CXX, "-x", "c++", "-std=c++17", "-O1", "-fPIC", "-shared", "-fopenmp", "-w", "-I", HERE,
           "-o", OUT] + srcs
    subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    print(build())

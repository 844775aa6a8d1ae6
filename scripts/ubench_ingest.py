#!/usr/bin/env python
"""Per-CU ingest rate on MI355X: LDS-DMA vs plain 16-byte loads, HBM-cold vs L2-hot, 64 .. 2048 workgroups.
(diagnostic behind DESIGN.md's tile-size reasoning for the deep-K convolutions)   python scripts/ubench_ingest.py"""
import ctypes
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(HERE, "ubench", "ingest.so")


def build():
    src = os.path.join(HERE, "ubench", "ingest.hip")
    if not os.path.exists(SO) or os.path.getmtime(SO) < os.path.getmtime(src):
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-o", SO, src])
    return SO


if __name__ == "__main__":
    build()
    if len(sys.argv) > 1 and sys.argv[1] == "build":
        sys.exit(0)
    import torch
    L = ctypes.CDLL(SO)
    L.ubench_ingest.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                ctypes.c_void_p, ctypes.c_void_p]
    dev = torch.device("cuda:0")
    buf = torch.randint(0, 255, (2 << 30,), dtype=torch.uint8, device=dev)  # 2 GiB
    sink = torch.zeros(4, dtype=torch.int32, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    print(f"{'mode':6s} {'source':28s} {'WGs':>5s} {'us':>9s} {'TB/s':>7s} {'GB/s per busy CU':>17s}")
    for mode, mname in ((0, "glds"), (1, "vgpr")):
        for src, span, nreg_of in (("HBM private regions", 4 << 20, lambda n: min(n, 512)),
                                   ("L2-hot, private 256 KB", 256 << 10, lambda n: n),
                                   ("L2-hot, ONE shared 2 MB", 2 << 20, lambda n: 1)):
            for nwg in (64, 128, 256, 512, 1024, 2048):
                iters = 256 if src.startswith("HBM") else 512
                nreg = nreg_of(nwg)
                for _ in range(2):
                    L.ubench_ingest(mode, buf.data_ptr(), span, nreg, iters, nwg, sink.data_ptr(), st)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(3):
                    L.ubench_ingest(mode, buf.data_ptr(), span, nreg, iters, nwg, sink.data_ptr(), st)
                e1.record()
                torch.cuda.synchronize()
                us = e0.elapsed_time(e1) / 3 * 1e3
                tot = nwg * iters * 16384
                print(f"{mname:6s} {src:28s} {nwg:5d} {us:9.1f} {tot / us / 1e6:7.2f} {tot / us / 1e3 / min(nwg, 256):17.1f}", flush=True)

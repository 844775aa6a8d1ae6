#!/usr/bin/env python
"""Does global_load_lds_dwordx4 accept source addresses that are only 2-byte aligned, and at what rate?"""
import ctypes, os, subprocess, sys
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import torch  # (first: the probe library must bind to the HIP runtime torch has loaded)
import ubench_ingest
L = ctypes.CDLL(ubench_ingest.build())
L.ubench_glds_misaligned.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
L.ubench_ingest.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
dev = torch.device("cuda:0")
src = torch.arange(4096, dtype=torch.int16, device=dev)
for off in (0, 2, 4, 6, 8, 14):
    out = torch.zeros(256, dtype=torch.int32, device=dev)
    rc = L.ubench_glds_misaligned(src.data_ptr(), off, out.data_ptr(), None)
    if rc: print("launch rc", rc)
    torch.cuda.synchronize()
    got = out.view(torch.int16)[:512].cpu()
    want = torch.arange(512, dtype=torch.int16) + off // 2
    print(f"offset {off:2d} bytes: {'EXACT' if torch.equal(got, want) else 'WRONG'}  first elements {got[:10].tolist()}")
buf = torch.randint(0, 255, (1 << 30,), dtype=torch.uint8, device=dev)
sink = torch.zeros(4, dtype=torch.int32, device=dev)
st = torch.cuda.current_stream().cuda_stream
for mode, name in ((0, "glds aligned"), (2, "glds 2-byte aligned")):
    for nwg in (256, 1024):
        for _ in range(2):
            L.ubench_ingest(mode, buf.data_ptr(), 4 << 20, min(nwg, 200), 256, nwg, sink.data_ptr(), st)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            L.ubench_ingest(mode, buf.data_ptr(), 4 << 20, min(nwg, 200), 256, nwg, sink.data_ptr(), st)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 3 * 1e3
        print(f"{name:22s} {nwg:5d} WGs  {nwg * 256 * 16384 / us / 1e6:6.2f} TB/s")

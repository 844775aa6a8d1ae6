#!/usr/bin/env python
"""Print the lane/element mapping of ds_read_b64_tr_b16 on this GPU (LDS element i holds the value i)."""
import ctypes
import os
import torch
HERE = os.path.dirname(os.path.abspath(__file__))
L = ctypes.CDLL(os.path.join(HERE, "ubench", "trprobe.so"))
L.trprobe_run.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
dev = torch.device("cuda:0")
RS = 100  # row stride in elements (a multiple of 4)
# within each 16-lane group gg: lane L points at row (4*gg + L//4), columns 4*(L%4) .. +3  -> value = row*RS + col
lane_elem = torch.tensor([(4 * (l // 16) + (l % 16) // 4) * RS + 4 * (l % 4) for l in range(64)], dtype=torch.int32, device=dev)
out = torch.zeros(256, dtype=torch.int16, device=dev)
assert L.trprobe_run(lane_elem.data_ptr(), out.data_ptr(), None) == 0
torch.cuda.synchronize()
o = out.cpu().view(64, 4).tolist()
for l in range(64):
    print(l, [(v // RS, v % RS) for v in o[l]])

#!/usr/bin/env python
"""Round-3 hardware probes (scripts/ubench/probe3.hip): 2-byte-aligned wide LDS reads; SGPR-base form of the LDS-DMA copy."""
import ctypes
import os
import torch
HERE = os.path.dirname(os.path.abspath(__file__))
L = ctypes.CDLL(os.path.join(HERE, "ubench", "probe3.so"))
P = ctypes.c_void_p
L.probe_lds_misaligned.argtypes = [P, P, P, P]
L.probe_glds_saddr.argtypes = [P, P, ctypes.c_int, P, P]
dev = torch.device("cuda:0")
for name, offs in (("aligned16", [l * 16 for l in range(64)]), ("aligned8", [8 + l * 16 for l in range(64)]),
                   ("aligned4", [4 + l * 24 for l in range(64)]), ("aligned2 (rows of 98 B)", [l * 98 for l in range(64)]),
                   ("aligned2+6", [6 + l * 98 for l in range(64)])):
    lb = torch.tensor(offs, dtype=torch.int32, device=dev)
    o64 = torch.zeros(128, dtype=torch.int32, device=dev)
    o128 = torch.zeros(256, dtype=torch.int32, device=dev)
    rc = L.probe_lds_misaligned(lb.data_ptr(), o64.data_ptr(), o128.data_ptr(), None)
    torch.cuda.synchronize()
    a = o64.cpu().view(torch.int16).view(64, 4).to(torch.int32) & 0xffff
    b = o128.cpu().view(torch.int16).view(64, 8).to(torch.int32) & 0xffff
    want64 = torch.tensor([[o // 2 + e for e in range(4)] for o in offs])
    want128 = torch.tensor([[o // 2 + e for e in range(8)] for o in offs])
    print(f"lds {name:26s} rc={rc} b64 ok={bool((a == want64).all())} b128 ok={bool((b == want128).all())}"
          f"  lane1 b64={a[1].tolist()} b128={b[1].tolist()} want={want128[1].tolist()}")
src = torch.arange(0, 32768, dtype=torch.int32, device=dev).to(torch.int16)
for name, base, offs in (("aligned", 1024, [l * 16 for l in range(64)]), ("2-byte aligned src", 1027, [l * 98 for l in range(64)]),
                         ("scattered", 64, [((l * 37) % 64) * 130 for l in range(64)])):
    lo = torch.tensor(offs, dtype=torch.int32, device=dev)
    out = torch.zeros(512, dtype=torch.int16, device=dev)
    rc = L.probe_glds_saddr(src.data_ptr(), lo.data_ptr(), base, out.data_ptr(), None)
    torch.cuda.synchronize()
    got = out.cpu().to(torch.int32).view(64, 8) & 0xffff
    want = torch.tensor([[(base + o // 2 + e) & 0xffff for e in range(8)] for o in offs])
    print(f"glds saddr {name:22s} rc={rc} ok={bool((got == want).all())} lane1 got={got[1].tolist()} want={want[1].tolist()}")

#!/usr/bin/env python
"""1x1 forward and data gradient through the C ABI under cot_set_tuning(23, value) against fp32 matmuls on the same bf16 operands
(device check for kernel forms that are not the default).   python scripts/check_fwd_variant.py 8"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from cotnet_amd import _lib  # noqa: E402

tune = int(sys.argv[1]) if len(sys.argv) > 1 else 0
L = _lib.lib()
assert L.cot_set_tuning(23, tune) == 0
dev = torch.device("cuda:0")
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
for (N, Ci, Co, H, bias) in [(80, 1024, 256, 14, False), (80, 2048, 512, 7, False), (80, 512, 2048, 7, False), (80, 256, 1024, 14, False),
                             (80, 128, 288, 14, True), (7, 320, 48, 7, False), (3, 64, 136, 14, True), (80, 512, 512, 7, False)]:
    torch.manual_seed(Ci + H)
    HW = H * H
    x = torch.randn(N, Ci, HW, device=dev).bfloat16()
    w = (torch.randn(Co, Ci, device=dev) / Ci ** 0.5).bfloat16()
    b = torch.randn(Co, device=dev).bfloat16() if bias else None
    y = torch.full((N, Co, HW), float("nan"), device=dev).bfloat16()
    assert L.cot_conv1x1_forward(P(x), None, Ci, P(w), P(b), P(y), N, Ci, Co, HW, _lib.COT_BF16, st) == 0, L.cot_last_error()
    ref = torch.einsum("oc,ncp->nop", w.float(), x.float()) + (b.float()[None, :, None] if bias else 0)
    ef = ((y.float() - ref).abs().max() / ref.abs().max()).item()
    gy = torch.randn(N, Co, HW, device=dev).bfloat16()
    gx = torch.full((N, Ci, HW), float("nan"), device=dev).bfloat16()
    ws = torch.empty(int(L.cot_conv1x1_workspace(N, Ci, Co, HW, 0)), dtype=torch.uint8, device=dev)
    assert L.cot_conv1x1_backward_data(P(gy), P(w), P(gx), None, Ci, 0, P(ws), N, Ci, Co, HW, _lib.COT_BF16, st) == 0
    refg = torch.einsum("oc,nop->ncp", w.float(), gy.float())
    eg = ((gx.float() - refg).abs().max() / refg.abs().max()).item()
    torch.cuda.synchronize()
    print(f"tune {tune}: N={N} {Ci}->{Co} @{H}: fwd rel err {ef:.2e}, dgrad {eg:.2e}")
    assert ef < 1.5e-2 and eg < 1.5e-2
print(f"tune {tune}: OK")

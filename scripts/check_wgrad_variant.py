#!/usr/bin/env python
"""Weight gradient of the 1x1 convolution through the C ABI under a given cot_set_tuning(25, ...) value against an fp32
einsum on the same bf16 operands, at CoTNet-50's deep shapes (device check for A/B variants that are not the default).

    python scripts/check_wgrad_variant.py 128"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from cotnet_amd import _lib  # noqa: E402


def main():
    tune = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    L = _lib.lib()
    assert L.cot_set_tuning(25, tune) == 0
    dev = torch.device("cuda:0")
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
    worst = 0.0
    for (N, Ci, Co, H, bias) in [(80, 1024, 256, 14, False), (80, 2048, 512, 7, False), (80, 512, 128, 28, False), (16, 256, 64, 56, False),
                                 (80, 128, 288, 14, True), (80, 256, 576, 7, True), (7, 96, 80, 7, False), (3, 200, 136, 12, True), (80, 64, 256, 56, False)]:
        torch.manual_seed(N + Ci)
        HW = H * H
        x = torch.randn(N, Ci, HW, device=dev).bfloat16()
        gy = torch.randn(N, Co, HW, device=dev).bfloat16()
        ref = torch.einsum("nop,ncp->oc", gy.float(), x.float())
        gw = torch.full((Co, Ci), float("nan"), device=dev).bfloat16()
        gb = torch.full((Co,), float("nan"), device=dev).bfloat16() if bias else None
        ws = torch.empty(int(L.cot_conv1x1_workspace(N, Ci, Co, HW, 1 if bias else 0)), dtype=torch.uint8, device=dev)
        rc = L.cot_conv1x1_backward_weight(P(gy), P(x), None, Ci, P(gw), P(gb), P(ws), N, Ci, Co, HW, _lib.COT_BF16, st)
        assert rc == 0, L.cot_last_error()
        torch.cuda.synchronize()
        e = ((gw.float() - ref).abs().max() / ref.abs().max()).item()
        eb = 0.0
        if bias:
            rb = gy.float().sum((0, 2))
            eb = ((gb.float() - rb).abs().max() / rb.abs().max()).item()
        worst = max(worst, e, eb)
        print(f"tune {tune}: N={N} {Ci}->{Co} @{H} bias={bias}: rel err {e:.2e} bias {eb:.2e} kernel {_lib.last_kernel()}")
        assert e < 1e-2 and eb < 1e-2
    print(f"tune {tune}: worst {worst:.2e} OK")


if __name__ == "__main__":
    main()

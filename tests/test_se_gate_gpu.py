"""The sigmoid gate of SplitAttnConv2d(radix=1) (csrc/radix_tail.hip: cot_se_gap / cot_se_gate / cot_se_gate_backward) on the
GPU: op level against the torch formula, and the whole module as SE-CoTNetD builds it (models/cotnet_hybrid.py:143-146,
models/layers/split_attn.py:62-88) against the plain module path."""
import copy

import pytest
import torch
from torch import nn

from cotnet_amd import conv3x3g as c3, fused_bn, radix_tail, se_gate
from cotnet_amd.layers import SplitAttnConv2d
from tests import truth

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("B,C,H", [(8, 64, 80), (8, 128, 40), (4, 256, 20), (3, 24, 7), (2, 8, 1)])
def test_gate_ops_match_torch(B, C, H, dtype):
    torch.manual_seed(C + H)
    x = torch.randn(B, C, H, H, device=DEV).to(dtype).requires_grad_(True)
    logits = (2 * torch.randn(B, C, device=DEV)).to(dtype).requires_grad_(True)
    g = torch.randn(B, C, H, H, device=DEV).to(dtype)
    assert se_gate.eligible(x)
    gap = se_gate.se_gap(x)
    out = se_gate.se_gate(x, logits)
    (out * g.to(out.dtype)).sum().backward()
    xr, lr = x.detach().double().requires_grad_(True), logits.detach().double().requires_grad_(True)
    outr = xr * torch.sigmoid(lr)[:, :, None, None]
    (outr * g.double()).sum().backward()
    tol = 1e-5 if dtype == torch.float32 else 2e-2
    assert torch.allclose(gap.double().flatten(1), xr.detach().mean((2, 3)), atol=tol, rtol=tol)
    assert torch.allclose(out.double(), outr.detach(), atol=tol, rtol=tol)
    assert torch.allclose(x.grad.double(), xr.grad, atol=tol, rtol=tol)
    scale = lr.grad.abs().max().item()
    assert (logits.grad.double() - lr.grad).abs().max().item() <= (2e-5 if dtype == torch.float32 else 2e-2) * scale + tol


def test_split_attn_module_fp32_matches_the_plain_path():
    torch.manual_seed(3)
    mod = SplitAttnConv2d(64, 64, 3, padding=1, radix=1, norm_layer=nn.BatchNorm2d).to(DEV).train()
    for m in mod.modules():
        if isinstance(m, nn.BatchNorm2d):
            nn.init.uniform_(m.weight, 0.5, 1.5)
            nn.init.uniform_(m.bias, -0.3, 0.3)
    ref = copy.deepcopy(mod).double()
    x = torch.randn(8, 64, 40, 40, device=DEV)
    g = torch.randn(8, 64, 40, 40, device=DEV)
    xi = x.clone().requires_grad_(True)
    with truth.switches(conv3x3="hip", fused_bn=True, fused_tail=True):
        y = mod(xi)
        assert "SeGate" in type(y.grad_fn).__name__
        y.backward(g)
    xr = x.double().requires_grad_(True)
    with truth.switches(**truth.PLAIN):
        yr = ref(xr)
        yr.backward(g.double())
    assert (y.double() - yr).abs().max() < 1e-3 and (xi.grad.double() - xr.grad).abs().max() < 1e-3   # BASELINE's fp32 bar
    for (n, p), (_, q) in zip(mod.named_parameters(), ref.named_parameters()):
        assert (p.grad.double() - q.grad).abs().max() <= 1e-3 * max(1.0, q.grad.abs().max().item()), n


def test_split_attn_module_mixed_bf16_against_the_fp32_truth():
    from cotnet_amd.flat_sgd import to_mixed_bf16
    torch.manual_seed(5)
    mod = SplitAttnConv2d(128, 128, 3, padding=1, radix=1, norm_layer=nn.BatchNorm2d).to(DEV).train()
    for m in mod.modules():
        if isinstance(m, nn.BatchNorm2d):
            nn.init.uniform_(m.weight, 0.5, 1.5)
            nn.init.uniform_(m.bias, -0.3, 0.3)
    to_mixed_bf16(mod)
    x = torch.randn(8, 128, 40, 40, device=DEV).bfloat16()
    g = torch.randn(8, 128, 40, 40, device=DEV).bfloat16()
    report = truth.check_against_truth(mod, x, g, cand=truth.ALL_HIP)
    assert report["y"][0] < 0.05, report

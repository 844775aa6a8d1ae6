"""Radix-2 tail (csrc/radix_tail.hip) on the GPU against the reference's 5-D formula (models/cotnet.py:92-104)."""
import pytest
import torch
from torch import nn

from cotnet_amd import radix_tail
from cotnet_amd.cotnet import radix2_fuse

pytestmark = pytest.mark.gpu
DEV = "cuda"


def reference_tail(x, k, se):
    B, C, H, W = x.shape
    x5 = torch.cat([x.view(B, C, 1, H, W), k.view(B, C, 1, H, W)], dim=2)
    gap = x5.sum(dim=2).mean((2, 3), keepdim=True)
    attn = torch.softmax(se(gap).view(B, C, 2), dim=2)
    return (x5 * attn.reshape(B, C, 2, 1, 1)).sum(dim=2).contiguous()


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("B,C,H", [(8, 64, 56), (8, 128, 28), (8, 256, 14), (8, 512, 7), (3, 24, 5)])
def test_matches_reference_formula(B, C, H, dtype):
    torch.manual_seed(C + H)
    A = max(C * 2 // 4, 32)
    se = nn.Sequential(nn.Conv2d(C, A, 1), nn.BatchNorm2d(A), nn.ReLU(inplace=True), nn.Conv2d(A, 2 * C, 1)).to(DEV).eval()
    if dtype == torch.bfloat16:
        for m in se:
            if isinstance(m, nn.Conv2d):
                m.to(torch.bfloat16)
    x = torch.randn(B, C, H, H, device=DEV).to(dtype)
    k = torch.randn(B, C, H, H, device=DEV).to(dtype)
    g = torch.randn(B, C, H, H, device=DEV).to(dtype)
    xa, ka = x.clone().requires_grad_(True), k.clone().requires_grad_(True)
    assert radix_tail.eligible(xa, ka)
    ya = radix2_fuse(xa, ka, se)
    ya.backward(g)
    ga = [p.grad.clone() for p in se.parameters()]
    se.zero_grad()
    xb, kb = x.clone().requires_grad_(True), k.clone().requires_grad_(True)
    yb = reference_tail(xb, kb, se)
    yb.backward(g)
    tol = 2e-5 if dtype == torch.float32 else 4e-2
    assert ((ya.float() - yb.float()).abs() <= tol * (1 + yb.float().abs())).all()
    assert ((xa.grad.float() - xb.grad.float()).abs() <= tol * (1 + xb.grad.float().abs())).all()
    assert ((ka.grad.float() - kb.grad.float()).abs() <= tol * (1 + kb.grad.float().abs())).all()
    for a, p in zip(ga, se.parameters()):
        if p.grad is not None:
            scale = p.grad.float().abs().max() + 1e-6
            assert (a.float() - p.grad.float()).abs().max() <= (tol * 20) * scale + tol

"""Classifier head on the library's kernels (cotnet_amd/head_fused.py, opt-in COT_HEAD=hip) against pool + nn.Linear."""
import pytest
import torch
import torch.nn.functional as F

from cotnet_amd import head_fused as hf
from cotnet_amd.layers import create_classifier

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.mark.parametrize("N,C,O,H", [(80, 2048, 1000, 7), (16, 2048, 1000, 10), (5, 64, 24, 7), (3, 128, 8, 1)])
def test_matches_pool_and_linear(N, C, O, H, monkeypatch):
    monkeypatch.setattr(hf, "MODE", "hip")
    torch.manual_seed(N)
    pool, fc = create_classifier(C, O, pool_type="avg")
    fc = fc.to(DEV).bfloat16()
    x = torch.randn(N, C, H, H, device=DEV).bfloat16().requires_grad_(True)
    assert hf.eligible(pool, fc, x)
    y = hf.head(pool, fc, x)
    assert y.shape == (N, O)
    g = torch.randn(N, O, device=DEV).bfloat16()
    y.backward(g)
    xr = x.detach().float().requires_grad_(True)
    wr, br = fc.weight.detach().float().requires_grad_(True), fc.bias.detach().float().requires_grad_(True)
    yr = F.linear(xr.mean((2, 3)), wr, br)
    yr.backward(g.float())
    assert torch.allclose(y.float(), yr.detach(), atol=2e-2, rtol=2e-2)
    assert (x.grad.float() - xr.grad).abs().max() <= 2e-2 * xr.grad.abs().max() + 1e-6
    assert (fc.weight.grad.float() - wr.grad).abs().max() <= 2e-2 * wr.grad.abs().max() + 1e-3
    assert (fc.bias.grad.float() - br.grad).abs().max() <= 2e-2 * br.grad.abs().max() + 1e-3


def test_other_heads_keep_the_modules(monkeypatch):
    monkeypatch.setattr(hf, "MODE", "hip")
    pool, fc = create_classifier(64, 10, pool_type="avg")   # 10 classes: not a multiple of 8
    fc = fc.to(DEV).bfloat16()
    x = torch.randn(4, 64, 7, 7, device=DEV).bfloat16()
    assert not hf.eligible(pool, fc, x)
    assert torch.equal(hf.head(pool, fc, x), fc(pool(x)))

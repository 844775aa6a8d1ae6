"""conv1x1-as-batched-GEMM (cotnet_amd/conv1x1.py, opt-in) computes exactly what nn.Conv2d computes."""
import pytest
import torch
from torch import nn

from cotnet_amd import conv1x1 as c1


@pytest.mark.parametrize("bias", [False, True])
def test_matches_conv2d(bias, monkeypatch):
    monkeypatch.setattr(c1, "MODE", "matmul")
    torch.manual_seed(0)
    conv = nn.Conv2d(12, 20, 1, bias=bias).double()
    x = torch.randn(3, 12, 5, 7, dtype=torch.float64, requires_grad=True)
    g = torch.randn(3, 20, 5, 7, dtype=torch.float64)
    assert c1.eligible(conv, x)
    y = c1.conv1x1(conv, x)
    y.backward(g)
    got = (y.detach().clone(), x.grad.clone(), conv.weight.grad.clone(), conv.bias.grad.clone() if bias else None)
    x.grad = None
    conv.zero_grad()
    yr = conv(x)
    yr.backward(g)
    assert torch.allclose(got[0], yr, atol=1e-12) and torch.allclose(got[1], x.grad, atol=1e-12)
    assert torch.allclose(got[2], conv.weight.grad, atol=1e-11)
    if bias:
        assert torch.allclose(got[3], conv.bias.grad, atol=1e-11)


def test_other_convolutions_fall_through(monkeypatch):
    monkeypatch.setattr(c1, "MODE", "matmul")
    x = torch.randn(2, 8, 6, 6)
    for conv in (nn.Conv2d(8, 8, 3, padding=1), nn.Conv2d(8, 8, 1, groups=2), nn.Conv2d(8, 8, 1, stride=2)):
        assert not c1.eligible(conv, x)
        assert torch.equal(c1.conv1x1(conv, x), conv(x))
    monkeypatch.setattr(c1, "MODE", "")
    assert not c1.eligible(nn.Conv2d(8, 8, 1), x)

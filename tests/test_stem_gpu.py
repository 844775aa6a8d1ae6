"""csrc/stem7x7.hip (opt-in COT_STEM=hip) on the GPU against nn.Conv2d(3, 64, 7, stride=2, padding=3) in fp32."""
import pytest
import torch
import torch.nn.functional as F
from torch import nn

from cotnet_amd import stem7x7 as s7

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.mark.parametrize("N,H", [(16, 224), (4, 256), (4, 320), (2, 64), (3, 32)])
def test_matches_torch_convolution(N, H, monkeypatch):
    monkeypatch.setattr(s7, "MODE", "hip")
    torch.manual_seed(H)
    conv = nn.Conv2d(3, 64, 7, stride=2, padding=3, bias=False).to(DEV).bfloat16()
    x = torch.randn(N, 3, H, H, device=DEV).bfloat16()
    assert s7.eligible(conv, x)
    y = s7.stem_conv(conv, x)
    gy = torch.randn_like(y)
    y.backward(gy)
    wr = conv.weight.detach().float().requires_grad_(True)
    yr = F.conv2d(x.float(), wr, None, 2, 3)
    yr.backward(gy.float())
    assert ((y.float() - yr).abs() <= 1e-2 * (yr.abs() + yr.abs().mean())).all()
    assert (conv.weight.grad.float() - wr.grad).abs().max() <= 1e-2 * wr.grad.abs().max()


def test_weight_gradient_is_deterministic_and_other_inputs_keep_the_module(monkeypatch):
    monkeypatch.setattr(s7, "MODE", "hip")
    conv = nn.Conv2d(3, 64, 7, stride=2, padding=3, bias=False).to(DEV).bfloat16()
    x = torch.randn(8, 3, 224, 224, device=DEV).bfloat16()
    grads = []
    for _ in range(2):
        conv.weight.grad = None
        y = s7.stem_conv(conv, x)
        y.backward(torch.ones_like(y))
        grads.append(conv.weight.grad.clone())
    assert torch.equal(grads[0], grads[1])
    assert not s7.eligible(conv, x.clone().requires_grad_(True))               # an input that needs a gradient
    assert not s7.eligible(conv, torch.zeros(2, 3, 30, 30, device=DEV).bfloat16())   # output width 15
    assert not s7.eligible(conv.float(), x.float())

"""csrc/group_norm9.hip (opt-in COT_GN9=hip) on the GPU against torch's GroupNorm evaluated in fp32."""
import pytest
import torch
import torch.nn.functional as F
from torch import nn

from cotnet_amd import group_norm9 as g9

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.mark.parametrize("N,dim,H", [(8, 64, 56), (8, 128, 28), (8, 256, 14), (8, 512, 7), (4, 256, 40), (3, 64, 80),
                                     (2, 64, 5)])
def test_matches_torch_group_norm(N, dim, H, monkeypatch):
    monkeypatch.setattr(g9, "MODE", "hip")
    torch.manual_seed(dim + H)
    C, G = 9 * dim // 8, dim // 8
    gn = nn.GroupNorm(G, C).to(DEV).bfloat16()
    with torch.no_grad():
        gn.weight.copy_(1 + 0.3 * torch.randn(C))
        gn.bias.copy_(0.2 * torch.randn(C))
    x = (torch.randn(N, C, H, H, device=DEV) * 1.5 + 0.4).bfloat16().requires_grad_(True)
    dy = torch.randn(N, C, H, H, device=DEV).bfloat16()
    assert g9.eligible(gn, x)
    y = g9.group_norm9(gn, x)
    y.backward(dy)
    xr = x.detach().float().requires_grad_(True)
    wr, br = gn.weight.detach().float().requires_grad_(True), gn.bias.detach().float().requires_grad_(True)
    yr = F.group_norm(xr, G, wr, br, gn.eps)
    yr.backward(dy.float())
    assert torch.allclose(y.float(), yr.detach(), atol=2e-2, rtol=2e-2)
    assert torch.allclose(x.grad.float(), xr.grad, atol=3e-2, rtol=3e-2)
    assert (gn.weight.grad.float() - wr.grad).abs().max() <= 1e-2 * wr.grad.abs().max() + 1e-2
    assert (gn.bias.grad.float() - br.grad).abs().max() <= 1e-2 * br.grad.abs().max() + 1e-2


def test_other_inputs_keep_the_module(monkeypatch):
    monkeypatch.setattr(g9, "MODE", "hip")
    gn = nn.GroupNorm(4, 36).to(DEV)
    x = torch.randn(2, 36, 8, 8, device=DEV)
    assert g9.eligible(gn, x)                                       # fp32: the fp32 kernels
    assert torch.allclose(g9.group_norm9(gn, x), gn(x), atol=1e-5, rtol=1e-5)
    assert not g9.eligible(gn.half(), x.half())                     # fp16: the module
    assert torch.equal(g9.group_norm9(gn, x.half()), gn(x.half()))
    gn = gn.float()
    assert not g9.eligible(nn.GroupNorm(4, 32).to(DEV).bfloat16(), torch.zeros(2, 32, 8, 8, device=DEV).bfloat16())

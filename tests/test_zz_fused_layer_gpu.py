"""CotLayer as one autograd node (cotnet_amd/cot_layer_fused.py, opt-in COT_FUSED_LAYER=1) on the GPU against the same
layer evaluated node-per-op with the same kernels (COT_CONV1X1=hip, COT_CONV3X3=hip): identical arithmetic and rounding
points except that dx / dk are summed in fp32 inside the data-gradient kernels.  (Sorts last: newest code.)"""
import copy

import pytest
import torch

from cotnet_amd import conv1x1 as c1, conv3x3g as c3, cot_layer_fused as clf
from cotnet_amd.cotnet import Bottleneck, CotLayer
from cotnet_amd.flat_sgd import to_mixed_bf16

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _rel(a, b):
    return ((a.float() - b.float()).abs().max() / (b.float().abs().max() + 1e-6)).item()


@pytest.mark.parametrize("N,C,H", [(8, 64, 56), (8, 128, 28), (8, 256, 14), (8, 512, 7)])
def test_single_node_layer_matches_node_per_op(N, C, H, monkeypatch):
    monkeypatch.setattr(c1, "MODE", "hip")
    monkeypatch.setattr(c3, "MODE", "hip")
    torch.manual_seed(C)
    node = CotLayer(C, 3).to(DEV).train()
    with torch.no_grad():
        for p in node.parameters():
            if p.ndim == 1:
                p.add_(0.3 * torch.randn_like(p))
    node = to_mixed_bf16(node)
    perop = copy.deepcopy(node)
    x = torch.randn(N, C, H, H, device=DEV).bfloat16()
    g = torch.randn(N, C, H, H, device=DEV).bfloat16()

    monkeypatch.setattr(clf, "ENABLED", False)
    xr = x.clone().requires_grad_(True)
    yr = perop(xr)
    yr.backward(g)
    monkeypatch.setattr(clf, "ENABLED", True)
    xf = x.clone().requires_grad_(True)
    assert clf.eligible(node, xf)
    yf = node(xf)
    assert yf.grad_fn.name().startswith("_CotLayerNode")
    yf.backward(g)
    torch.cuda.synchronize()

    assert _rel(yf, yr.detach()) < 1e-2
    assert _rel(xf.grad, xr.grad) < 2e-2
    pr = dict(perop.named_parameters())
    top = max(q.grad.float().abs().max() for q in pr.values())
    for n_, p in node.named_parameters():
        assert p.grad is not None and p.grad.shape == p.shape and p.grad.dtype == p.dtype, n_
        if pr[n_].grad.float().abs().max() > 1e-3 * top:
            assert _rel(p.grad, pr[n_].grad) < 3e-2, (n_, _rel(p.grad, pr[n_].grad))
    br, bf = dict(perop.named_buffers()), dict(node.named_buffers())
    for n_ in br:
        assert torch.allclose(bf[n_].float(), br[n_].float(), atol=1e-3, rtol=1e-3), n_


def test_ineligible_inputs_take_the_ordinary_forward(monkeypatch):
    monkeypatch.setattr(clf, "ENABLED", True)
    layer = to_mixed_bf16(CotLayer(64, 3).to(DEV))
    x = torch.randn(2, 64, 8, 8, device=DEV).bfloat16()
    assert clf.eligible(layer.train(), x)
    assert not clf.eligible(layer.eval(), x)                                  # inference
    assert not clf.eligible(layer.train(), x.to(memory_format=torch.channels_last))
    assert not clf.eligible(CotLayer(64, 3).to(DEV).train(), x.float())       # fp32 model
    y = layer.eval()(x)
    assert y.shape == x.shape


def test_bottleneck_trains_with_the_single_node_layer(monkeypatch):
    """two SGD steps of a Bottleneck with the fused layer node inside: loss decreases, parameters stay finite"""
    monkeypatch.setattr(clf, "ENABLED", True)
    monkeypatch.setattr(c1, "MODE", "hip")
    monkeypatch.setattr(c3, "MODE", "hip")
    torch.manual_seed(0)
    blk = Bottleneck(256, 64).to(DEV).train()
    with torch.no_grad():
        blk.bn3.weight.fill_(1.0)
    blk = to_mixed_bf16(blk)
    x = torch.randn(8, 256, 28, 28, device=DEV).bfloat16()
    tgt = torch.randn(8, 256, 28, 28, device=DEV).bfloat16()
    opt = torch.optim.SGD(blk.parameters(), lr=1e-2)
    losses = []
    for _ in range(3):
        opt.zero_grad()
        loss = (blk(x).float() - tgt.float()).square().mean()
        loss.backward()
        opt.step()
        losses.append(loss.item())
    assert losses[-1] < losses[0]
    assert all(torch.isfinite(p.float()).all() for p in blk.parameters())

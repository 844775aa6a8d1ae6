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


@pytest.mark.parametrize("kind", ["identity", "project", "stride2"])
def test_single_node_bottleneck_matches_node_per_op(kind, monkeypatch):
    """the whole Bottleneck as one node (identity shortcut / 1x1 projection / stride-2 block with avd pooling) against the
    node-per-op path on the same kernels; ReLU masks may flip on bf16 ulps, so gradients are compared in the mean"""
    from cotnet_amd import pool3x3 as p3
    from cotnet_amd.resnet import downsample_conv
    monkeypatch.setattr(c1, "MODE", "hip")
    monkeypatch.setattr(c3, "MODE", "hip")
    monkeypatch.setattr(p3, "MODE", "hip")
    torch.manual_seed(7)
    stride = 2 if kind == "stride2" else 1
    inpl = 256 if kind == "identity" else 128
    ds = None if kind == "identity" else downsample_conv(inpl, 256, 1, stride=stride)
    node = Bottleneck(inpl, 64, stride=stride, downsample=ds).to(DEV).train()
    with torch.no_grad():
        node.bn3.weight.fill_(0.8)
    node = to_mixed_bf16(node)
    perop = copy.deepcopy(node)
    x = torch.randn(8, inpl, 28, 28, device=DEV).bfloat16()
    g = torch.randn(8, 256, 28 // stride, 28 // stride, device=DEV).bfloat16()

    def run(blk, enabled):
        monkeypatch.setattr(clf, "ENABLED", enabled)
        xi = x.clone().requires_grad_(True)
        y = blk(xi)
        y.backward(g)
        return y, xi.grad, {n: p.grad for n, p in blk.named_parameters()}

    yr, gxr, gr = run(perop, False)
    yf, gxf, gf = run(node, True)
    assert yf.grad_fn.name().startswith("_BottleneckNode") and not yr.grad_fn.name().startswith("_BottleneckNode")

    def mrel(a, b):
        return ((a.float() - b.float()).abs().mean() / (b.float().abs().mean() + 1e-9)).item()

    assert mrel(yf, yr) < 1e-2
    assert mrel(gxf, gxr) < 6e-2
    top = max(v.float().abs().max() for v in gr.values())
    for n_, v in gf.items():
        if gr[n_].float().abs().max() > 1e-3 * top and not n_.endswith("se.0.bias"):
            assert mrel(v, gr[n_]) < 0.12, (n_, mrel(v, gr[n_]))

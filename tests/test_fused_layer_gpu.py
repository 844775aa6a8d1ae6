"""CotLayer / Bottleneck as ONE autograd node (cotnet_amd/cot_layer_fused.py, opt-in COT_FUSED_LAYER=1) on the GPU.
Each composed result is compared with an fp32 evaluation of the same module (tests/truth.py): the single-node path must
not sit further from that truth than the round-1 path (MIOpen convolutions, one node per op) does."""
import copy

import pytest
import torch

from cotnet_amd import cot_layer_fused as clf
from cotnet_amd.cotnet import Bottleneck, CotLayer
from cotnet_amd.flat_sgd import to_mixed_bf16
from tests import truth

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.mark.parametrize("N,C,H", [(8, 64, 56), (8, 128, 28), (8, 256, 14), (8, 512, 7)])
def test_single_node_layer_against_fp32_truth(N, C, H):
    torch.manual_seed(C)
    layer = CotLayer(C, 3).to(DEV).train()
    with torch.no_grad():
        for p in layer.parameters():
            if p.ndim == 1:
                p.add_(0.3 * torch.randn_like(p))
    layer = to_mixed_bf16(layer)
    x = torch.randn(N, C, H, H, device=DEV).bfloat16()
    g = torch.randn(N, C, H, H, device=DEV).bfloat16()
    truth.check_against_truth(layer, x, g, cand=truth.SINGLE_NODE)
    # the node really is the single one, it returns a gradient for every parameter, and the BatchNorm buffers move as
    # torch's do
    _, _, gf, mf, node = truth.run(layer, x, g, want_module=True, **truth.SINGLE_NODE)
    assert node.startswith("_CotLayerNode")
    _, _, gr, mr, node_r = truth.run(layer, x, g, want_module=True, **truth.ROUND1)
    assert not node_r.startswith("_CotLayerNode")
    assert set(gf) == set(gr) == {n for n, _ in layer.named_parameters()}
    # running statistics: against the fp32 truth's buffers (bf16 activations move a batch variance by ~0.5 %, i.e. the
    # running variance by ~5e-4 after one momentum-0.1 update: both paths must stay in that class)
    *_, mt, _ = truth.run(copy.deepcopy(layer).float(), x.float(), g.float(), want_module=True, **truth.PLAIN)
    for (n_, a), (_, b), (_, tb) in zip(mf.named_buffers(), mr.named_buffers(), mt.named_buffers()):
        ea, eb = (a.float() - tb.float()).abs().max().item(), (b.float() - tb.float()).abs().max().item()
        assert ea <= 2.0 * eb + 2e-3 * (1 + tb.float().abs().max().item()), (n_, ea, eb)


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


def test_bottleneck_trains_with_the_single_node_layer():
    """two SGD steps of a Bottleneck with the fused layer node inside: loss decreases, parameters stay finite"""
    torch.manual_seed(0)
    blk = Bottleneck(256, 64).to(DEV).train()
    with torch.no_grad():
        blk.bn3.weight.fill_(1.0)
    blk = to_mixed_bf16(blk)
    x = torch.randn(8, 256, 28, 28, device=DEV).bfloat16()
    tgt = torch.randn(8, 256, 28, 28, device=DEV).bfloat16()
    opt = torch.optim.SGD(blk.parameters(), lr=1e-2)
    losses = []
    with truth.switches(**truth.SINGLE_NODE):
        for _ in range(3):
            opt.zero_grad()
            loss = (blk(x).float() - tgt.float()).square().mean()
            loss.backward()
            opt.step()
            losses.append(loss.item())
    assert losses[-1] < losses[0]
    assert all(torch.isfinite(p.float()).all() for p in blk.parameters())


@pytest.mark.parametrize("kind", ["identity", "project", "stride2"])
def test_single_node_bottleneck_against_fp32_truth(kind):
    """the whole Bottleneck as one node (identity shortcut / 1x1 projection / stride-2 block with avd pooling)"""
    from cotnet_amd.resnet import downsample_conv
    torch.manual_seed(7)
    stride = 2 if kind == "stride2" else 1
    inpl = 256 if kind == "identity" else 128
    ds = None if kind == "identity" else downsample_conv(inpl, 256, 1, stride=stride)
    blk = Bottleneck(inpl, 64, stride=stride, downsample=ds).to(DEV).train()
    with torch.no_grad():
        blk.bn3.weight.fill_(0.8)
    blk = to_mixed_bf16(blk)
    x = torch.randn(8, inpl, 28, 28, device=DEV).bfloat16()
    g = torch.randn(8, 256, 28 // stride, 28 // stride, device=DEV).bfloat16()
    truth.check_against_truth(blk, x, g, cand=truth.SINGLE_NODE)
    *_, node = truth.run(blk, x, g, want_module=True, **truth.SINGLE_NODE)
    assert node.startswith("_BottleneckNode")

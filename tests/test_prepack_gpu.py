"""Grouped 3x3 weights packed ahead of time (cot_layer_fused.after_optimizer_step): a few training steps of a small
CoTNet stage with FlatSGD, the packings made on the side stream after every optimizer step, against the same steps with the packing
inline (COT_PREPACK off).  The packing is the same bytes either way, so losses and final weights must be IDENTICAL -- a stale or
half-written packing (a missing stream dependency, a key that misses a parameter update) shows as a difference."""
import copy

import pytest
import torch
from torch import nn

from cotnet_amd import cot_layer_fused as clf
from cotnet_amd.cotnet import Bottleneck
from cotnet_amd.flat_sgd import FlatSGD, to_mixed_bf16
from cotnet_amd.resnet import downsample_conv
from tests import truth

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _run(stage, x, steps, prepack):
    m = copy.deepcopy(stage)
    old = clf.PREPACK
    clf.PREPACK = prepack
    for k in list(clf._PACKS.keys()):
        del clf._PACKS[k]
    clf._PACK_EVENT.clear()
    hits = []
    try:
        with truth.switches(**truth.SINGLE_NODE):
            opt = FlatSGD(m, lr=0.05, momentum=0.9, weight_decay=1e-4)
            losses = []
            for i in range(steps):
                opt.zero_grad()
                loss = m(x).float().square().mean()
                loss.backward()
                opt.step()
                losses.append(loss.detach().clone())
                hits.append(sum(1 for e in clf._PACKS.values() if 0 in e and 1 in e))
            torch.cuda.synchronize()
    finally:
        clf.PREPACK = old
    return torch.stack(losses), [p.detach().float().clone() for p in m.parameters()], hits


def test_prepacked_training_steps_equal_inline_packing():
    torch.manual_seed(3)
    stage = nn.Sequential(Bottleneck(128, 64, stride=2, downsample=downsample_conv(128, 256, 1, stride=2)), Bottleneck(256, 64),
                          Bottleneck(256, 64)).to(DEV).train()
    with torch.no_grad():
        for b in stage:
            b.bn3.weight.fill_(0.8)
    stage = to_mixed_bf16(stage)
    x = torch.randn(16, 128, 28, 28, device=DEV).bfloat16()
    la, pa, hits = _run(stage, x, 4, prepack=True)
    lb, pb, _ = _run(stage, x, 4, prepack=False)
    assert hits[0] == 3 and hits[-1] == 3, hits       # every layer's two packings exist after the first optimizer step
    assert torch.equal(la, lb), (la, lb)
    for a, b in zip(pa, pb):
        assert torch.equal(a, b)
    assert la[-1] < la[0]                              # (and the steps train)


def test_a_parameter_update_behind_torchs_back_invalidates_the_packing():
    """the flat SGD kernel writes weights through raw pointers (torch's version counter does not move): PARAM_EPOCH is the key's part that
    notices; an in-place torch update moves the version counter; both must send the next forward back to inline packing"""
    torch.manual_seed(4)
    blk = to_mixed_bf16(Bottleneck(256, 64).to(DEV).train())
    x = torch.randn(16, 256, 14, 14, device=DEV).bfloat16()
    conv = blk.conv2.key_embed[0]
    with truth.switches(**truth.SINGLE_NODE):
        opt = FlatSGD(blk, lr=0.01)
        for _ in range(2):
            opt.zero_grad()
            blk(x).float().mean().backward()
            opt.step()
        L = __import__("cotnet_amd._lib", fromlist=["lib"]).lib()
        g = clf._PACKS[conv]["geom"]
        assert clf._pack_for(L, conv, 0, *g) is not None
        clf.PARAM_EPOCH[0] += 1                       # "somebody's kernel changed the weights"
        assert clf._pack_for(L, conv, 0, *g) is None
        clf.PARAM_EPOCH[0] -= 1
        assert clf._pack_for(L, conv, 0, *g) is not None
        with torch.no_grad():
            conv.weight.mul_(1.0)                     # an in-place torch op: the version counter moves
        assert clf._pack_for(L, conv, 0, *g) is None
    torch.cuda.synchronize()


def test_raw_data_writes_and_graph_replays_go_through_invalidate_packs():
    """`p.data.copy_()` (a broadcast of the module's state, an EMA swap-in) and a replayed HIP graph that contains the optimizer move
    neither torch's version counter nor the optimizer epoch: cot_layer_fused.invalidate_packs() is the public call that makes the next
    forward pack inline again -- and the step after such a write computes on the NEW weights"""
    torch.manual_seed(5)
    blk = to_mixed_bf16(Bottleneck(256, 64).to(DEV).train())
    x = torch.randn(16, 256, 14, 14, device=DEV).bfloat16()
    conv = blk.conv2.key_embed[0]
    with truth.switches(**truth.SINGLE_NODE):
        opt = FlatSGD(blk, lr=0.01)
        for _ in range(2):
            opt.zero_grad()
            blk(x).float().mean().backward()
            opt.step()
        L = __import__("cotnet_amd._lib", fromlist=["lib"]).lib()
        g = clf._PACKS[conv]["geom"]
        assert clf._pack_for(L, conv, 0, *g) is not None
        new_w = (conv.weight.detach().float() * 0.5).bfloat16()
        conv.weight.data.copy_(new_w)                 # version counter of the Parameter's .data alias: torch does move it here ...
        ref = copy.deepcopy(blk)
        clf.invalidate_packs()                        # ... but the contract does not rely on it
        assert clf._pack_for(L, conv, 0, *g) is None
        with torch.no_grad():
            y = blk(x)
        old = clf.PREPACK
        clf.PREPACK = False
        try:
            with torch.no_grad():
                y_ref = ref(x)
        finally:
            clf.PREPACK = old
        assert torch.equal(y, y_ref)
    torch.cuda.synchronize()

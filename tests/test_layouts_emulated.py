"""The per-tensor-layout kernels (cot_*_lay) on the host emulator: tests/layout_cases.py with CPU tensors."""
import ctypes

import pytest

from cotnet_amd import _lib
from tests import layout_cases as lc
from tests.emul import build_emul

try:
    _EMUL = ctypes.CDLL(build_emul.build())
    for _name, (_res, _args) in _lib.SYMBOLS.items():
        getattr(_EMUL, _name).restype = _res
        getattr(_EMUL, _name).argtypes = _args
except FileNotFoundError:
    _EMUL = None

pytestmark = pytest.mark.skipif(_EMUL is None, reason="host emulation build unavailable")


@pytest.fixture(autouse=True)
def _knobs():
    _EMUL.cot_set_tuning(18, 256)
    _EMUL.cot_set_tuning(21, 1)  # channel-resident BatchNorm kernels on
    yield


@pytest.mark.parametrize("HW", [196, 49, 64])
@pytest.mark.parametrize("act,res,y2,ps", [(1, False, True, False), (0, False, False, False), (2, False, False, False), (1, True, False, True)])
def test_batchnorm_forward_layouts(HW, act, res, y2, ps):
    lc.bn_forward_case(_EMUL, "cpu", None, 6, 8, HW, act, res, y2, ps)


@pytest.mark.parametrize("HW", [196, 49, 64])
@pytest.mark.parametrize("act,res,dy2,ps", [(1, False, True, False), (0, False, False, False), (2, False, False, False), (1, True, False, True)])
def test_batchnorm_backward_layouts(HW, act, res, dy2, ps):
    lc.bn_backward_case(_EMUL, "cpu", None, 6, 8, HW, act, res, dy2, ps)


@pytest.mark.parametrize("HW", [196, 49])
def test_radix_tail_layouts(HW):
    lc.radix_case(_EMUL, "cpu", None, 5, 16, HW)


@pytest.mark.parametrize("HW", [196, 49, 784])
def test_group_norm9_layouts(HW):
    lc.gn9_case(_EMUL, "cpu", None, 3, 4, HW)


@pytest.mark.parametrize("H,blocks,coxt", [(14, 3, False), (7, 2, False), (14, 2, True)])
def test_channel_major_bottlenecks_against_the_nchw_single_node(H, blocks, coxt, monkeypatch):
    """a run of identity Bottlenecks of a deep stage (models/cotnet.py:181-264) through cot_layer_fused._BottleneckCMNode -- first block
    NCHW in / channel-major out, middle ones channel-major both sides, last one back to NCHW -- against the same blocks through the
    NCHW single-node path (_BottleneckNode), both on the host-emulated kernels.  Same kernels and rounding points; what differs is the
    order of a few sums (channel rows of N*HW elements instead of N rows of HW), so the two agree to bf16 ulps, with a few flipped ReLU
    decisions at bn3 (compared like tests/test_kernels_emulated.py::test_fused_bottleneck_node_on_emulated_kernels)."""
    import copy

    import torch
    from torch import nn

    from cotnet_amd import cot_layer_fused as clf, conv1x1 as c1, conv3x3g as c3, fused_bn, radix_tail
    from cotnet_amd.cotnet import Bottleneck
    from cotnet_amd.flat_sgd import to_mixed_bf16
    torch.manual_seed(6 + H)
    N, W = 6 if H == 14 else 8, H
    # (coxt: CoTNeXt's block -- cardinality 2, base width 48 -> CoXtLayer(96 * planes / 64): grouped 1x1s, interleaved [x, k], group -> batch fold)
    stage = nn.Sequential(*[Bottleneck(512, 128, cardinality=2, base_width=48).train() if coxt else Bottleneck(256, 64).train()
                            for _ in range(blocks)])
    assert type(stage[0].conv2).__name__ == ("CoXtLayer" if coxt else "CotLayer")
    with torch.no_grad():
        for p in stage.parameters():
            if p.ndim == 1:
                p.add_(0.3 * torch.randn_like(p))
        for b in stage:
            b.bn3.weight.fill_(0.8)
    stage = to_mixed_bf16(stage)
    ref = copy.deepcopy(stage)
    inpl = 512 if coxt else 256
    x = torch.randn(N, inpl, H, W).bfloat16()
    g = torch.randn(N, inpl, H, W).bfloat16()
    monkeypatch.setattr(_lib, "lib", lambda: _EMUL)
    for mod in (clf, c1, c3, fused_bn, radix_tail):
        monkeypatch.setattr(mod, "_DEVICE_ONLY", False)
    caches = (clf._SIZES, clf._MASKS, clf._BSIZES, clf._CM_SIZES, clf._CM_OK)
    for cache in caches:
        cache.clear()
    monkeypatch.setattr(clf, "ENABLED", True)

    monkeypatch.setattr(clf, "CM_LAYOUT", False)
    xr = x.clone().requires_grad_(True)
    yr = ref(xr)
    assert yr.grad_fn.name().startswith("_BottleneckNode")
    yr.backward(g)

    monkeypatch.setattr(clf, "CM_LAYOUT", True)
    clf.plan_stage_layouts(stage)
    assert [b._next_cm for b in stage] == [True] * (blocks - 1) + [False]
    xf = x.clone().requires_grad_(True)
    h, seen = xf, []
    for b in stage:
        assert clf.cm_block_eligible(b, h)
        h = b(h)
        assert h.grad_fn.name().startswith("_BottleneckCMNode")
        seen.append(clf._is_cm(h))
    assert seen == [True] * (blocks - 1) + [False] and h.is_contiguous()
    h.backward(g)

    def relmax(a, b):
        return ((a.float() - b.float()).abs().max() / (b.float().abs().max() + 1e-6)).item()

    def rel(a, b):
        return ((a.float() - b.float()).abs().mean() / (b.float().abs().mean() + 1e-6)).item()

    assert relmax(h, yr.detach()) < 1e-2
    assert rel(xf.grad, xr.grad) < 6e-2 and xf.grad.is_contiguous()
    pr = dict(ref.named_parameters())
    top = max(q.grad.float().abs().max() for q in pr.values())
    for n_, p in stage.named_parameters():
        assert p.grad is not None and p.grad.shape == p.shape and p.grad.dtype == p.dtype, n_
        if pr[n_].grad.float().abs().max() > 1e-3 * top and not n_.endswith("se.0.bias"):
            assert rel(p.grad, pr[n_].grad) < 0.12, (n_, rel(p.grad, pr[n_].grad))
    br, bf = dict(ref.named_buffers()), dict(stage.named_buffers())
    for n_ in br:
        assert torch.allclose(bf[n_].float(), br[n_].float(), atol=1e-3, rtol=1e-3), n_
    for cache in caches:
        cache.clear()


@pytest.mark.parametrize("N,Ci,Co,HW,act,res,mask", [(3, 64, 64, 784, 1, False, False), (2, 64, 256, 3136, 1, True, True), (2, 128, 32, 784, 0, False, False)])
def test_batchnorm_statistics_from_the_convolution_epilogue(N, Ci, Co, HW, act, res, mask):
    lc.bn_epilogue_case(_EMUL, "cpu", None, N, Ci, Co, HW, act, res, mask)


@pytest.mark.parametrize("kind,H", [("identity", 6), ("stride2", 12), ("project", 28)])
def test_eval_mode_bottleneck_as_one_call_sequence(kind, H, monkeypatch):
    """inference (BASELINE config 2): cot_layer_fused.eval_block_forward -- the block's kernels back to back on the running statistics,
    autograd off -- against the module's ordinary eval forward (node per op, same emulated kernels where they are hand-written)"""
    import copy

    import torch

    import cotnet_amd.aggregation_zeropad as az
    from cotnet_amd import cot_layer_fused as clf, conv1x1 as c1, conv3x3g as c3, fused_bn, pool3x3 as p3, radix_tail
    from cotnet_amd.cotnet import Bottleneck
    from cotnet_amd.flat_sgd import to_mixed_bf16
    from cotnet_amd.resnet import downsample_conv
    from tests.test_kernels_emulated import _EmulAggregation
    torch.manual_seed(21 + H)
    stride = 2 if kind == "stride2" else 1
    inpl = 256 if kind == "identity" else 128
    ds = None if kind == "identity" else downsample_conv(inpl, 256, 1, stride=stride)
    blk = Bottleneck(inpl, 64, stride=stride, downsample=ds)
    with torch.no_grad():
        for m in blk.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.running_mean.normal_(0, 0.3)
                m.running_var.uniform_(0.5, 1.5)
                m.weight.uniform_(0.5, 1.5)
                m.bias.normal_(0, 0.2)
    blk = to_mixed_bf16(blk).eval()
    x = torch.randn(2, inpl, H, H).bfloat16()
    monkeypatch.setattr(_lib, "lib", lambda: _EMUL)
    for mod in (clf, c1, c3, fused_bn, radix_tail, p3):
        monkeypatch.setattr(mod, "_DEVICE_ONLY", False)
    monkeypatch.setattr(az, "aggregation_zeropad", lambda i, w, kernel_size=3, stride=1, padding=0, dilation=1: _EmulAggregation.apply(i, w))
    for cache in (clf._SIZES, clf._MASKS, clf._BSIZES):
        cache.clear()
    with torch.no_grad():
        monkeypatch.setattr(clf, "ENABLED", False)
        ref = copy.deepcopy(blk)(x).float()
        monkeypatch.setattr(clf, "ENABLED", True)
        assert clf.eval_block_eligible(blk, x)
        clf.reset_node_counts()
        y = blk(x)
        assert clf.NODE_COUNTS["bottleneck_eval"] == 1 and y.grad_fn is None
    assert not clf.eval_block_eligible(blk, x)  # (autograd on again: the ordinary path)
    assert ((y.float() - ref).abs() <= 2e-2 * (ref.abs() + ref.abs().mean())).all(), (y.float() - ref).abs().max()
    for cache in (clf._SIZES, clf._MASKS, clf._BSIZES):
        cache.clear()


def test_channel_major_stage_with_its_opening_block(monkeypatch):
    """a whole deep stage -- the stride-2 opening block (avd pooling in front of the layer, stride-2 projection shortcut; models/cotnet.py:
    228-264, models/resnet.py:364-394) followed by identity blocks -- on the channel-major node: the opening block takes NCHW, runs its
    layer / conv3 / bn3 / projection BatchNorm channel-major and hands a channel-major tensor on; against the NCHW single nodes"""
    import copy

    import torch
    from torch import nn

    from cotnet_amd import cot_layer_fused as clf, conv1x1 as c1, conv3x3g as c3, fused_bn, pool3x3 as p3, radix_tail
    from cotnet_amd.cotnet import Bottleneck
    from cotnet_amd.flat_sgd import to_mixed_bf16
    from cotnet_amd.resnet import downsample_conv
    torch.manual_seed(31)
    N, H = 6, 28
    stage = nn.Sequential(Bottleneck(128, 64, stride=2, downsample=downsample_conv(128, 256, 1, stride=2)).train(),
                          Bottleneck(256, 64).train(), Bottleneck(256, 64).train())
    assert stage[0].avd is not None
    with torch.no_grad():
        for p in stage.parameters():
            if p.ndim == 1:
                p.add_(0.3 * torch.randn_like(p))
        for b in stage:
            b.bn3.weight.fill_(0.8)
    stage = to_mixed_bf16(stage)
    ref = copy.deepcopy(stage)
    x = torch.randn(N, 128, H, H).bfloat16()
    g = torch.randn(N, 256, H // 2, H // 2).bfloat16()
    monkeypatch.setattr(_lib, "lib", lambda: _EMUL)
    for mod in (clf, c1, c3, fused_bn, radix_tail, p3):
        monkeypatch.setattr(mod, "_DEVICE_ONLY", False)
    caches = (clf._SIZES, clf._MASKS, clf._BSIZES, clf._CM_SIZES, clf._CM_OK)
    for cache in caches:
        cache.clear()
    monkeypatch.setattr(clf, "ENABLED", True)
    monkeypatch.setattr(clf, "CM_LAYOUT", False)
    xr = x.clone().requires_grad_(True)
    yr = ref(xr)
    yr.backward(g)
    monkeypatch.setattr(clf, "CM_LAYOUT", True)
    clf.plan_stage_layouts(stage)
    assert [b._next_cm for b in stage] == [True, True, False]
    xf = x.clone().requires_grad_(True)
    h, seen = xf, []
    for b in stage:
        assert clf.cm_block_eligible(b, h)
        h = b(h)
        assert h.grad_fn.name().startswith("_BottleneckCMNode")
        seen.append(clf._is_cm(h))
    assert seen == [True, True, False]
    h.backward(g)

    def rel(a, b):
        return ((a.float() - b.float()).abs().mean() / (b.float().abs().mean() + 1e-6)).item()

    assert ((h.float() - yr.detach().float()).abs().max() / yr.detach().float().abs().max()).item() < 1e-2
    assert rel(xf.grad, xr.grad) < 6e-2 and xf.grad.is_contiguous()
    pr = dict(ref.named_parameters())
    top = max(q.grad.float().abs().max() for q in pr.values())
    for n_, p in stage.named_parameters():
        assert p.grad is not None and p.grad.shape == p.shape, n_
        if pr[n_].grad.float().abs().max() > 1e-3 * top and not n_.endswith("se.0.bias"):
            assert rel(p.grad, pr[n_].grad) < 0.12, (n_, rel(p.grad, pr[n_].grad))
    br, bf = dict(ref.named_buffers()), dict(stage.named_buffers())
    for n_ in br:
        assert torch.allclose(bf[n_].float(), br[n_].float(), atol=1e-3, rtol=1e-3), n_
    for cache in caches:
        cache.clear()


@pytest.mark.parametrize("N,C,G,H", [(3, 256, 4, 14), (2, 512, 4, 7), (2, 64, 4, 56), (2, 128, 4, 28), (2, 384, 8, 14)])
def test_conv3x3_on_weights_packed_ahead_of_time(N, C, G, H):
    """cot_conv3x3g_pack + cot_conv3x3g_forward_packed / _backward_data_packed against the ordinary entry points (which pack per call):
    bit-identical -- the key embedding of every stage (models/cotnet.py:43-47) and CoTNeXt's groups-8 form"""
    import torch
    L, P, BF = _EMUL, lc.P, lc.BF
    torch.manual_seed(C + H)
    x, gy = torch.randn(N, C, H, H).bfloat16(), torch.randn(N, C, H, H).bfloat16()
    w = (torch.randn(C, C // G, 3, 3) / (9 * C // G) ** 0.5).bfloat16()
    masks = torch.empty(int(L.cot_conv3x3g_masks_bytes(H, H)), dtype=torch.uint8)
    assert L.cot_conv3x3g_masks(P(masks), H, H, None) == 0
    ws = torch.empty(int(L.cot_conv3x3g_workspace(N, C, C, G, H, H)), dtype=torch.uint8)
    y0, y1 = torch.full_like(x, float("nan")), torch.full_like(x, float("nan"))
    gx0, gx1 = torch.full_like(x, float("nan")), torch.full_like(x, float("nan"))
    assert L.cot_conv3x3g_forward(P(x), P(w), P(y0), P(masks), P(ws), N, C, C, G, H, H, BF, None) == 0
    assert L.cot_conv3x3g_backward_data(P(gy), P(w), P(gx0), 0, P(masks), P(ws), N, C, C, G, H, H, BF, None) == 0
    nb = int(L.cot_conv3x3g_packed_bytes(C, C, G))
    pf, pd = torch.full((nb,), 0xEE, dtype=torch.uint8), torch.full((nb,), 0xEE, dtype=torch.uint8)
    assert L.cot_conv3x3g_pack(P(w), P(pf), 0, N, C, C, G, H, H, BF, None) == 0, L.cot_last_error()
    assert L.cot_conv3x3g_pack(P(w), P(pd), 1, N, C, C, G, H, H, BF, None) == 0, L.cot_last_error()
    assert L.cot_conv3x3g_forward_packed(P(x), P(pf), P(y1), N, C, C, G, H, H, BF, None) == 0, L.cot_last_error()
    assert L.cot_conv3x3g_backward_data_packed(P(gy), P(pd), P(gx1), 0, N, C, C, G, H, H, BF, None) == 0, L.cot_last_error()
    assert torch.equal(y0, y1) and torch.equal(gx0, gx1)
    acc = gx0.clone()
    assert L.cot_conv3x3g_backward_data_packed(P(gy), P(pd), P(acc), 1, N, C, C, G, H, H, BF, None) == 0
    ref = torch.full_like(x, float("nan"))
    ref.copy_(gx0)
    assert L.cot_conv3x3g_backward_data(P(gy), P(w), P(ref), 1, P(masks), P(ws), N, C, C, G, H, H, BF, None) == 0
    assert torch.equal(acc, ref)

"""The general grouped convolution kernels (csrc/conv_gen.hip) on the GPU: fp32 tensors on the fp32 MFMA (the reference's
own precision, config.yaml `amp: False`), grouped 1x1 convolutions of CoXtLayer (models/cotnet.py:123-131) and grouped 3x3
convolutions with 12 / 24 channels per group (:112-116) in fp32 and bf16 -- against torch's convolution evaluated in fp64 on
the same operands.  Tolerances: fp32 2e-5 of the tensor's scale (accumulation order only), bf16 1e-2 (one rounding of an
fp32-accumulated sum)."""
import pytest
import torch
import torch.nn.functional as F
from torch import nn

from cotnet_amd import conv1x1 as c1, conv3x3g as c3, group_norm9 as g9
from tests import truth

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _close(a, b, dtype):
    rel = 2e-5 if dtype == torch.float32 else 1e-2
    return ((a.double() - b).abs() <= rel * (b.abs() + b.abs().mean())).all()


def _ref(conv, x, gy):
    """fp64 autograd reference on the same (possibly bf16-rounded) values"""
    xr = x.detach().double().requires_grad_(True)
    wr = conv.weight.detach().double().requires_grad_(True)
    br = conv.bias.detach().double().requires_grad_(True) if conv.bias is not None else None
    y = F.conv2d(xr, wr, br, conv.stride, conv.padding, conv.dilation, conv.groups)
    y.backward(gy.double())
    return y.detach(), xr.grad, wr.grad, (br.grad if br is not None else None)


# (N, Ci, Co, groups, H, bias)
CASES_1X1 = [
    (4, 256, 64, 1, 56, False),    # Bottleneck.conv1, stage 1
    (4, 128, 32, 1, 56, False),    # CotLayer.embed[0] (on the concatenation)
    (4, 32, 72, 1, 56, True),      # CotLayer.embed[3]
    (3, 512, 2048, 1, 7, False),   # Bottleneck.conv3, stage 4
    (3, 2048, 512, 1, 7, False),   # Bottleneck.conv1, stage 4: 128 reduction steps
    (4, 192, 48, 2, 56, False),    # CoXtLayer.embed[0], dim 96: 96 -> 24 per group
    (4, 48, 108, 2, 56, True),     # CoXtLayer.embed[3]: 24 -> 54 per group
    (4, 96, 96, 2, 56, False),     # CoXtLayer.conv1x1[0]: 48 -> 48 per group
    (3, 768, 768, 2, 7, False),    # the same at dim 768
    (2, 10, 6, 2, 5, True),        # 5 -> 3 per group
]


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("N,Ci,Co,G,H,bias", CASES_1X1)
def test_conv1x1_matches_torch(N, Ci, Co, G, H, bias, dtype, monkeypatch):
    monkeypatch.setattr(c1, "MODE", "hip")
    torch.manual_seed(Ci + Co + H)
    conv = nn.Conv2d(Ci, Co, 1, groups=G, bias=bias).to(DEV).to(dtype)
    x = torch.randn(N, Ci, H, H, device=DEV).to(dtype).requires_grad_(True)
    gy = torch.randn(N, Co, H, H, device=DEV).to(dtype)
    if dtype == torch.bfloat16 and G == 1 and Ci % 8 == 0 and Co % 8 == 0:
        pytest.skip("bf16, groups 1, channels on the MFMA-32 grid: the tuned kernels' case (tests/test_conv1x1_gpu.py)")
    assert c1.eligible_general(conv, x)
    y = c1.conv1x1(conv, x)
    assert "Conv1x1" in type(y.grad_fn).__name__, type(y.grad_fn).__name__
    y.backward(gy)
    torch.cuda.synchronize()
    yr, gxr, gwr, gbr = _ref(conv, x, gy)
    assert _close(y, yr, dtype)
    assert _close(x.grad, gxr, dtype)
    assert _close(conv.weight.grad, gwr, dtype)
    if bias:
        assert _close(conv.bias.grad, gbr, dtype)


# (N, C, groups, H)
CASES_3X3 = [(4, 64, 4, 56, torch.float32), (4, 128, 4, 28, torch.float32), (3, 256, 4, 14, torch.float32),
             (3, 512, 4, 7, torch.float32),          # CotLayer.key_embed at the reference's precision
             (4, 96, 8, 56, torch.float32), (4, 96, 8, 56, torch.bfloat16),      # CoXtLayer.key_embed: 12 per group
             (4, 192, 8, 28, torch.bfloat16),                                      # 24 per group
             (2, 20, 2, 5, torch.float32), (1, 6, 1, 1, torch.bfloat16)]


@pytest.mark.parametrize("N,C,G,H,dtype", CASES_3X3)
def test_conv3x3_matches_torch(N, C, G, H, dtype, monkeypatch):
    monkeypatch.setattr(c3, "MODE", "hip")
    torch.manual_seed(C + H)
    conv = nn.Conv2d(C, C, 3, padding=1, groups=G, bias=False).to(DEV).to(dtype)
    x = torch.randn(N, C, H, H, device=DEV).to(dtype).requires_grad_(True)
    gy = torch.randn(N, C, H, H, device=DEV).to(dtype)
    assert c3.eligible(conv, x)
    y = c3.conv3x3(conv, x)
    assert "Conv3x3G" in type(y.grad_fn).__name__
    y.backward(gy)
    torch.cuda.synchronize()
    yr, gxr, gwr, _ = _ref(conv, x, gy)
    assert _close(y, yr, dtype)
    assert _close(x.grad, gxr, dtype)
    assert _close(conv.weight.grad, gwr, dtype)


def test_weight_gradients_are_deterministic(monkeypatch):
    monkeypatch.setattr(c1, "MODE", "hip")
    conv = nn.Conv2d(192, 48, 1, groups=2).to(DEV)
    x = torch.randn(8, 192, 28, 28, device=DEV, requires_grad=True)
    gy = torch.randn(8, 48, 28, 28, device=DEV)
    grads = []
    for _ in range(3):
        conv.zero_grad()
        c1.conv1x1(conv, x).backward(gy)
        grads.append((conv.weight.grad.clone(), conv.bias.grad.clone()))
    assert all(torch.equal(g[0], grads[0][0]) and torch.equal(g[1], grads[0][1]) for g in grads)


@pytest.mark.parametrize("N,dim,H", [(8, 64, 56), (8, 128, 28), (8, 256, 14), (8, 512, 7), (2, 96, 120)])
def test_group_norm9_fp32_matches_torch(N, dim, H, monkeypatch):
    monkeypatch.setattr(g9, "MODE", "hip")
    torch.manual_seed(dim)
    gn = nn.GroupNorm(dim // 8, 9 * dim // 8).to(DEV)
    nn.init.uniform_(gn.weight, 0.5, 1.5)
    nn.init.uniform_(gn.bias, -0.3, 0.3)
    x = (torch.randn(N, 9 * dim // 8, H, H, device=DEV) * 1.7 + 0.6).requires_grad_(True)
    gy = torch.randn_like(x)
    assert g9.eligible(gn, x)
    y = g9.group_norm9(gn, x)
    assert "GroupNorm9" in type(y.grad_fn).__name__
    y.backward(gy)
    xr = x.detach().double().requires_grad_(True)
    wr, br = gn.weight.detach().double().requires_grad_(True), gn.bias.detach().double().requires_grad_(True)
    yr = F.group_norm(xr, gn.num_groups, wr, br, gn.eps)
    yr.backward(gy.double())
    assert (y.double() - yr).abs().max() < 2e-5
    assert (x.grad.double() - xr.grad).abs().max() < 1e-4
    assert ((gn.weight.grad.double() - wr.grad).abs() <= 1e-4 * (wr.grad.abs() + wr.grad.abs().mean())).all()
    assert ((gn.bias.grad.double() - br.grad).abs() <= 1e-4 * (br.grad.abs() + br.grad.abs().mean())).all()


def test_coxt_layer_mixed_bf16_against_the_fp32_truth():
    """CoXtLayer in bench.py's mixed precision with every convolution and the GroupNorm on the library's kernels vs. the
    MIOpen / torch path, both measured against the fp32 evaluation (tests/truth.py)"""
    from cotnet_amd.cotnet import CoXtLayer
    from cotnet_amd.flat_sgd import to_mixed_bf16
    torch.manual_seed(11)
    layer = CoXtLayer(96, 3).to(DEV).train()
    for m in layer.modules():
        if isinstance(m, (nn.BatchNorm2d, nn.GroupNorm)):
            nn.init.uniform_(m.weight, 0.5, 1.5)
            nn.init.uniform_(m.bias, -0.3, 0.3)
    to_mixed_bf16(layer)
    x = torch.randn(8, 96, 28, 28, device=DEV).bfloat16()
    gy = torch.randn(8, 96, 28, 28, device=DEV).bfloat16()
    report = truth.check_against_truth(layer, x, gy, cand=truth.ALL_HIP)
    assert report["y"][0] < 0.05, report

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
    assert not s7.eligible(conv, x.float())                                    # (dtypes must agree; fp32 with fp32 is taken: below)


@pytest.mark.parametrize("N,H", [(8, 224), (2, 256), (2, 64), (3, 32)])
def test_fp32_stem_matches_torch_convolution(N, H, monkeypatch):
    """the stem at the reference's own precision (csrc/stem7x7_f32.hip) against torch's fp32 convolution; deterministic weight gradient"""
    monkeypatch.setattr(s7, "MODE", "hip")
    torch.manual_seed(H)
    torch.backends.cudnn.allow_tf32 = False
    conv = nn.Conv2d(3, 64, 7, stride=2, padding=3, bias=False).to(DEV)
    x = torch.randn(N, 3, H, H, device=DEV)
    assert s7.eligible(conv, x)
    y = s7.stem_conv(conv, x)
    gy = torch.randn_like(y)
    y.backward(gy)
    g1 = conv.weight.grad.clone()
    conv.weight.grad = None
    y2 = s7.stem_conv(conv, x)
    y2.backward(gy)
    assert torch.equal(y, y2) and torch.equal(g1, conv.weight.grad)
    wr = conv.weight.detach().double().requires_grad_(True)
    yr = F.conv2d(x.double(), wr, None, 2, 3)
    yr.backward(gy.double())
    assert (y.double() - yr).abs().max() <= 1e-5 * yr.abs().max()
    assert (g1.double() - wr.grad).abs().max() <= 2e-5 * wr.grad.abs().max()


def test_fp32_cotnet50_step_makes_no_module_fallback():
    """CoTNet-50 in fp32 (the reference's precision, cot_experiments/CoTNet-50-350epoch/config.yaml:2): forward + backward of the package
    default without one torch-module convolution / Linear (VERDICT r5 missing #4: stem, stride-2 projections, head)"""
    from cotnet_amd import _lib, create_model
    torch.manual_seed(3)
    model = create_model("cotnet50", num_classes=1000).to(DEV).train()
    x = torch.randn(2, 3, 224, 224, device=DEV)
    _lib.FALLBACKS.clear()
    loss = model(x).float().logsumexp(1).mean()
    loss.backward()
    torch.cuda.synchronize()
    assert not _lib.FALLBACKS, dict(_lib.FALLBACKS)
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in model.parameters())


# ---- a deep stem's three 3x3 convolutions (models/cotnet_hybrid.py:359-368): csrc/stem3x3.hip + conv_lds.hip, groups = 1 ----------
@pytest.mark.parametrize("N,H,Co", [(8, 320, 64), (4, 224, 32), (2, 64, 64), (3, 32, 32)])
def test_deep_stem_first_convolution_matches_torch(N, H, Co, monkeypatch):
    from cotnet_amd import stem3x3 as s3
    monkeypatch.setattr(s7, "MODE", "hip")
    torch.manual_seed(H + Co)
    conv = nn.Conv2d(3, Co, 3, stride=2, padding=1, bias=False).to(DEV).bfloat16()
    x = torch.randn(N, 3, H, H, device=DEV).bfloat16()
    assert s3.eligible(conv, x)
    y = s3.stem3x3_conv(conv, x)
    gy = torch.randn_like(y)
    y.backward(gy)
    g1 = conv.weight.grad.clone()
    wr = conv.weight.detach().float().requires_grad_(True)
    yr = F.conv2d(x.float(), wr, None, 2, 1)
    yr.backward(gy.float())
    assert ((y.float() - yr).abs() <= 2.0 ** -8 * yr.abs() + 1e-5).all()     # fp32 sums of 27 exact products, one bf16 rounding
    assert (g1.float() - wr.grad).abs().max() <= 1e-2 * wr.grad.abs().max()
    conv.weight.grad = None
    y2 = s3.stem3x3_conv(conv, x)
    y2.backward(gy)
    assert torch.equal(y, y2) and torch.equal(g1, conv.weight.grad)            # deterministic
    assert not s3.eligible(conv, x.clone().requires_grad_(True))
    assert not s3.eligible(conv, torch.zeros(2, 3, 30, 30, device=DEV).bfloat16())


def test_se_cotnetd_stem_makes_no_module_fallback(monkeypatch):
    """SE-CoTNetD's stem (stem_width 64) in the measured form -- bf16 weights, library kernels -- runs without a single module
    convolution and follows an fp32 evaluation of the same modules"""
    import copy
    from cotnet_amd import _lib, conv3x3g, resnet
    from cotnet_amd.flat_sgd import to_mixed_bf16
    monkeypatch.setattr(s7, "MODE", "hip")
    monkeypatch.setattr(conv3x3g, "MODE", "hip")
    torch.manual_seed(31)
    conv1, inplanes = resnet.make_stem(3, 64, "deep", nn.BatchNorm2d, nn.ReLU)
    stem = nn.ModuleDict(dict(conv1=conv1, bn1=nn.BatchNorm2d(inplanes), act1=nn.ReLU(inplace=True))).to(DEV).train()
    ref = copy.deepcopy(stem)
    stem = to_mixed_bf16(stem)
    x = torch.randn(4, 3, 320, 320, device=DEV).bfloat16()
    _lib.FALLBACKS.clear()
    y = resnet.stem_forward(stem["conv1"], stem["bn1"], stem["act1"], x)
    assert not _lib.FALLBACKS, dict(_lib.FALLBACKS)
    g = torch.randn_like(y)
    y.backward(g)
    for p_, q_ in zip(ref.parameters(), stem.parameters()):   # the same (rounded) weights in fp32
        p_.data.copy_(q_.data.float())
    yr = ref["act1"](ref["bn1"](ref["conv1"](x.float())))
    yr.backward(g.float())
    assert (y.float() - yr).norm() <= 2e-2 * yr.norm()
    for (n_, a), (_, b) in zip(stem.named_parameters(), ref.named_parameters()):
        assert (a.grad.float() - b.grad).norm() <= 0.1 * b.grad.norm() + 1e-3, n_

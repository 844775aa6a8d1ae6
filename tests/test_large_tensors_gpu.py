"""Maximum sizes: activations of more than 2^31 elements (N = 10752 images of 64 x 56 x 56; a 288 GB device holds dozens of them)
through the 1x1 convolution kernels and the fused BatchNorm, checked by properties that need no full-size reference: the last
images (the ones behind offset 2^31) against a small-batch evaluation, the weight gradient against the sum over two halves
(linearity), BatchNorm's backward identities sum(dx) = 0 and sum(dx * xhat) = 0 per channel.  The aggregation's own test of this kind
is tests/test_agg_gpu.py::test_tensors_beyond_2_31_elements_index_correctly."""
import pytest
import torch
import torch.nn.functional as F
from torch import nn

from cotnet_amd import conv1x1 as c1, fused_bn

pytestmark = pytest.mark.gpu
DEV = "cuda"
N, C, H = 10752, 64, 56


def _room(gb):
    torch.cuda.empty_cache()  # (blocks cached by earlier tests of the same process count as used otherwise)
    free, _ = torch.cuda.mem_get_info()
    if free < gb * 1e9:
        pytest.skip(f"needs ~{gb} GB of device memory")


def _near(a, b, rel=2e-2):
    b = b.float()
    return (a.float() - b).abs().max().item() <= rel * b.abs().max().item()


def test_conv1x1_beyond_2_31_elements(monkeypatch):
    assert N * C * H * H > 2 ** 31
    _room(60)
    monkeypatch.setattr(c1, "MODE", "hip")
    g = torch.Generator(device=DEV).manual_seed(5)
    conv = nn.Conv2d(C, C, 1, bias=False).to(DEV).bfloat16()
    x = torch.randn(N, C, H, H, device=DEV, dtype=torch.bfloat16, generator=g).requires_grad_(True)
    gy = torch.randn(N, C, H, H, device=DEV, dtype=torch.bfloat16, generator=g)
    assert c1.eligible_hip(conv, x)
    y = c1.conv1x1(conv, x)
    assert "Conv1x1" in type(y.grad_fn).__name__
    y.backward(gy)
    torch.cuda.synchronize()
    wf = conv.weight.detach().float()
    for sl in (slice(0, 2), slice(N - 2, N)):
        assert _near(y.detach()[sl], F.conv2d(x.detach()[sl].float(), wf)), sl
        assert _near(x.grad[sl], F.conv2d(gy[sl].float(), wf.transpose(0, 1).contiguous())), sl
    gw_full = conv.weight.grad.detach().float().clone()
    gw_sum = torch.zeros_like(gw_full)
    for sl in (slice(0, N // 2), slice(N // 2, N)):   # each half stays below 2^31 elements
        conv.weight.grad = None
        xs = x.detach()[sl].requires_grad_(True)
        c1.conv1x1(conv, xs).backward(gy[sl])
        gw_sum += conv.weight.grad.detach().float()
    torch.cuda.synchronize()
    assert _near(gw_full, gw_sum), (gw_full - gw_sum).abs().max().item()


def test_fused_batchnorm_beyond_2_31_elements():
    _room(90)
    g = torch.Generator(device=DEV).manual_seed(6)
    bn = nn.BatchNorm2d(C).to(DEV).train()
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5, generator=g)
        bn.bias.normal_(0, 0.2, generator=g)
    x = (torch.randn(N, C, H, H, device=DEV, dtype=torch.bfloat16, generator=g) * 1.5 + 0.5).requires_grad_(True)
    res = torch.randn(N, C, H, H, device=DEV, dtype=torch.bfloat16, generator=g).requires_grad_(True)
    dy = torch.randn(N, C, H, H, device=DEV, dtype=torch.bfloat16, generator=g)
    y = fused_bn.fused_bn_act(x, bn, "relu", res)
    assert "BNAct" in type(y.grad_fn).__name__
    y.backward(dy)
    torch.cuda.synchronize()
    var, mean = torch.var_mean(x.detach().float(), dim=(0, 2, 3), unbiased=False)
    assert torch.allclose(bn.running_mean, 0.1 * mean, atol=1e-4, rtol=1e-4)
    rstd = (var + bn.eps).rsqrt()
    a, b = (bn.weight * rstd).view(1, C, 1, 1), (bn.bias - mean * bn.weight * rstd).view(1, C, 1, 1)
    for sl in (slice(0, 2), slice(N - 2, N)):
        want = torch.relu(x.detach()[sl].float() * a + b + res.detach()[sl].float())
        assert (y.detach()[sl].float() - want).abs().max().item() <= 2e-2 * want.abs().max().item(), sl
        gz = dy[sl].float() * (y.detach()[sl] > 0)
        assert torch.equal(res.grad[sl].float(), gz.to(torch.bfloat16).float()), sl   # dresidual = dy * relu'
    # BatchNorm backward identities per channel (any upstream gradient): sum dx = 0, sum dx * xhat = 0 -- relative to sum |dx|
    dx = x.grad.float()
    xhat = (x.detach().float() - mean.view(1, C, 1, 1)) * rstd.view(1, C, 1, 1)
    scale = dx.abs().sum((0, 2, 3))
    assert ((dx.sum((0, 2, 3))).abs() <= 2e-3 * scale).all()
    assert (((dx * xhat).sum((0, 2, 3))).abs() <= 2e-3 * scale).all()
    gz = dy.float() * (y.detach() > 0)
    assert torch.allclose(bn.bias.grad, gz.sum((0, 2, 3)), rtol=1e-3, atol=1.0)
    assert torch.allclose(bn.weight.grad, (gz * xhat).sum((0, 2, 3)), rtol=1e-3, atol=1.0)


def test_grouped_conv3x3_beyond_2_31_elements(monkeypatch):
    """CotLayer.key_embed's convolution (64 channels, 4 groups, 56 x 56) on the same batch: last images against torch on the slice,
    weight gradient against the sum over two halves"""
    from cotnet_amd import conv3x3g as c3
    _room(60)
    monkeypatch.setattr(c3, "MODE", "hip")
    g = torch.Generator(device=DEV).manual_seed(7)
    conv = nn.Conv2d(C, C, 3, padding=1, groups=4, bias=False).to(DEV).bfloat16()
    x = torch.randn(N, C, H, H, device=DEV, dtype=torch.bfloat16, generator=g).requires_grad_(True)
    gy = torch.randn(N, C, H, H, device=DEV, dtype=torch.bfloat16, generator=g)
    assert c3.eligible(conv, x)
    y = c3.conv3x3(conv, x)
    assert "Conv3x3G" in type(y.grad_fn).__name__
    y.backward(gy)
    torch.cuda.synchronize()
    wf = conv.weight.detach().float()
    for sl in (slice(0, 2), slice(N - 2, N)):
        xs = x.detach()[sl].float().requires_grad_(True)
        ys = F.conv2d(xs, wf, None, 1, 1, 1, 4)
        ys.backward(gy[sl].float())
        assert _near(y.detach()[sl], ys.detach()), sl
        assert _near(x.grad[sl], xs.grad), sl
    gw_full = conv.weight.grad.detach().float().clone()
    gw_sum = torch.zeros_like(gw_full)
    for sl in (slice(0, N // 2), slice(N // 2, N)):
        conv.weight.grad = None
        xs = x.detach()[sl].requires_grad_(True)
        c3.conv3x3(conv, xs).backward(gy[sl])
        gw_sum += conv.weight.grad.detach().float()
    torch.cuda.synchronize()
    assert _near(gw_full, gw_sum), (gw_full - gw_sum).abs().max().item()

"""csrc/pool3x3.hip (opt-in COT_POOL=hip) on the GPU against nn.MaxPool2d(3, 2, 1) / nn.AvgPool2d(3, 2, padding=1)."""
import pytest
import torch
from torch import nn

from cotnet_amd import pool3x3 as p3

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("N,C,H", [(8, 64, 112), (8, 128, 56), (8, 256, 28), (8, 512, 14), (3, 5, 9), (2, 3, 1)])
@pytest.mark.parametrize("kind", ["max", "avg"])
def test_matches_torch_pooling(kind, N, C, H, dtype, monkeypatch):
    monkeypatch.setattr(p3, "MODE", "hip")
    torch.manual_seed(H)
    mod = nn.MaxPool2d(3, 2, 1) if kind == "max" else nn.AvgPool2d(3, 2, padding=1)
    x = torch.relu(torch.randn(N, C, H, H, device=DEV)).to(dtype)   # post-ReLU: ties at zero everywhere
    xa, xb = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
    assert p3.eligible(mod, xa)
    ya = p3.pool(mod, xa)
    yb = mod(xb)
    g = torch.randn_like(yb)
    ya.backward(g)
    yb.backward(g)
    if kind == "max":
        assert torch.equal(ya, yb) and torch.equal(xa.grad, xb.grad)   # same arg-max rule: bit-identical
    else:
        tol = 1e-5 if dtype == torch.float32 else 2e-2
        assert torch.allclose(ya.float(), yb.float(), atol=tol, rtol=tol)
        assert torch.allclose(xa.grad.float(), xb.grad.float(), atol=tol, rtol=tol)


def test_other_poolings_keep_the_module(monkeypatch):
    monkeypatch.setattr(p3, "MODE", "hip")
    x = torch.randn(2, 4, 8, 8, device=DEV)
    for mod in (nn.MaxPool2d(2, 2), nn.AvgPool2d(2, 1), nn.AvgPool2d(2, 2, padding=1), nn.MaxPool2d(3, 1, 1)):
        assert not p3.eligible(mod, x)
        assert torch.equal(p3.pool(mod, x), mod(x))
    odd = x[:, :, :7, :7].contiguous()  # (a 2 x 2 average pooling with ceil_mode on an odd plane has partial windows: the module)
    mod = nn.AvgPool2d(2, 2, ceil_mode=True, count_include_pad=False)
    assert not p3.eligible(mod, odd) and torch.equal(p3.pool(mod, odd), mod(odd))
    assert not p3.eligible(nn.MaxPool2d(3, 2, 1), x.double())


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("N,C,H", [(8, 64, 160), (8, 128, 80), (8, 256, 40), (8, 512, 20), (3, 5, 9), (2, 3, 2)])
def test_blur_pool_matches_the_reference_formula(N, C, H, dtype, monkeypatch):
    """BlurPool2d(filt_size=3, stride=2) of SE-CoTNetD (models/layers/blur_pool.py:53-58) on the stencil kernels vs. the
    module's own ReflectionPad2d + depthwise conv2d, at the tensor sizes of se_cotnetd_152_L at 320 x 320"""
    from cotnet_amd.layers import BlurPool2d
    torch.manual_seed(H)
    mod = BlurPool2d(C).to(DEV)
    x = torch.randn(N, C, H, H, device=DEV).to(dtype)
    xa, xb = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
    monkeypatch.setattr(p3, "MODE", "hip")
    assert p3.blur_eligible(xa)
    ya = mod(xa)
    assert "BlurPool" in type(ya.grad_fn).__name__
    monkeypatch.setattr(p3, "MODE", "")
    yb = mod(xb)
    g = torch.randn_like(yb)
    ya.backward(g)
    yb.backward(g)
    tol = 1e-5 if dtype == torch.float32 else 2e-2
    assert torch.allclose(ya.float(), yb.float(), atol=tol, rtol=tol)
    assert torch.allclose(xa.grad.float(), xb.grad.float(), atol=tol, rtol=tol)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("N,C,H", [(64, 256, 80), (64, 512, 40), (64, 1024, 20), (3, 5, 6), (2, 3, 2)])
def test_avgpool2x2_of_the_avg_down_shortcut_is_exact(N, C, H, dtype, monkeypatch):
    """nn.AvgPool2d(2, 2, ceil_mode=True, count_include_pad=False) of downsample_avg (models/resnet.py:380-394) at the tensor
    sizes of se_cotnetd_152_L at 320 x 320, B = 64: `pool()` takes the library's kernel, forward and gradient equal torch's
    bit for bit"""
    torch.manual_seed(H)
    mod = nn.AvgPool2d(2, 2, ceil_mode=True, count_include_pad=False)
    x = torch.randn(N, C, H, H, device=DEV).to(dtype)
    xa, xb = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
    monkeypatch.setattr(p3, "MODE", "hip")
    assert p3.eligible(mod, xa)
    ya = p3.pool(mod, xa)
    assert "AvgPool2" in type(ya.grad_fn).__name__
    yb = mod(xb)
    g = torch.randn_like(yb)
    ya.backward(g)
    yb.backward(g)
    assert torch.equal(ya, yb) and torch.equal(xa.grad, xb.grad)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("N,C,H", [(80, 256, 56), (80, 512, 28), (80, 1024, 14), (3, 5, 6), (2, 3, 2)])
def test_subsample2_is_exact(N, C, H, dtype):
    """cot_subsample2_*: the input of a stride-2 1x1 projection shortcut (x[:, :, ::2, ::2]) and its gradient, bit-exact; the
    gradient kernel writes every element (NaN-prefilled output)"""
    import ctypes
    from cotnet_amd import _lib
    L = _lib.lib()
    P = lambda t: ctypes.c_void_p(t.data_ptr())  # noqa: E731
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    torch.manual_seed(H)
    dt = _lib.dtype_code(dtype)
    x = torch.randn(N, C, H, H, device=DEV).to(dtype)
    y = torch.full((N, C, H // 2, H // 2), float("nan"), device=DEV).to(dtype)
    assert L.cot_subsample2_forward(P(x), P(y), N * C, H, H, dt, st) == 0, L.cot_last_error()
    assert torch.equal(y, x[:, :, ::2, ::2])
    gy = torch.randn_like(y)
    gx = torch.full_like(x, float("nan"))
    assert L.cot_subsample2_backward(P(gy), P(gx), N * C, H, H, dt, st) == 0, L.cot_last_error()
    ref = torch.zeros_like(x)
    ref[:, :, ::2, ::2] = gy
    assert torch.equal(gx, ref)


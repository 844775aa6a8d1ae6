"""The study kernels of the channels-last route on a real MI355X at the layer shapes of CoTNet-50's 14 x 14 / 7 x 7
stages (B = 80), against torch on the same bf16 operands.  The round that wrote them ended its GPU budget with the forward GEMM's
measurement (profiles/r04_layout_study_own_kc_gemm.log: results equal torch's), so the others have run on the host emulator only
(tests/test_kernels_emulated.py::test_*channels_last*, test_k_contiguous_*): this file is the first thing to run on the next GPU
session -- `COT_STUDY_GPU=1 python -m pytest tests/test_channels_last_study_gpu.py -m gpu` -- and is skipped without that variable so
that kernels off every model's path cannot colour the product's GPU suite."""
import ctypes
import os

import pytest
import torch
import torch.nn.functional as F

from cotnet_amd import _lib

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(os.environ.get("COT_STUDY_GPU", "0") != "1", reason="study kernels: set COT_STUDY_GPU=1")]
DEV = "cuda"
BF = 2


def P(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(None)


def S():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def near(a, b, rel=2e-2):
    b = b.float()
    return (a.float() - b).abs().max().item() <= rel * b.abs().max().item()


@pytest.mark.parametrize("HW,K,Nn,k1,bias", [(196, 1024, 256, 1024, False), (196, 256, 1024, 256, False), (49, 2048, 512, 2048, False),
                                             (196, 512, 128, 256, False), (196, 128, 288, 128, True), (49, 256, 576, 256, True)])
def test_conv1x1_forms(HW, K, Nn, k1, bias):
    L, M = _lib.lib(), 80 * HW
    torch.manual_seed(K + Nn)
    x, w = torch.randn(M, K, device=DEV).bfloat16(), (torch.randn(Nn, K, device=DEV) / K ** 0.5).bfloat16()
    b = torch.randn(Nn, device=DEV).bfloat16() if bias else None
    x1, x2 = x[:, :k1].contiguous(), (x[:, k1:].contiguous() if k1 < K else None)
    y = torch.full((M, Nn), float("nan"), device=DEV).bfloat16()
    assert L.cot_study_conv1x1_nhwc(P(x1), P(x2), k1, P(w), P(b), P(y), 0, M, Nn, K, 0, S()) == 0
    ref = x.float() @ w.float().t() + (b.float() if bias else 0)
    assert near(y, ref)
    L.cot_study_conv1x1_nhwc_wgrad_workspace.restype = ctypes.c_size_t
    dy = torch.randn(M, Nn, device=DEV).bfloat16()
    ws = torch.empty(L.cot_study_conv1x1_nhwc_wgrad_workspace(M, K, Nn, 0), device=DEV, dtype=torch.uint8)
    dw = torch.full((Nn, K), float("nan"), device=DEV).bfloat16()
    assert L.cot_study_conv1x1_nhwc_wgrad(P(x), P(dy), P(dw), P(ws), M, K, Nn, 0, S()) == 0
    assert near(dw, dy.float().t() @ x.float(), 1e-2)
    dx = torch.full((M, K), float("nan"), device=DEV).bfloat16()   # data gradient straight from the untransposed weight [Co][Ci]
    assert L.cot_study_conv1x1_nhwc_dgrad(P(dy), P(w), P(dx), 0, M, K, Nn, K, K, 0, S()) == 0
    assert near(dx, dy.float() @ w.float())


@pytest.mark.parametrize("H,D", [(14, 256), (7, 512)])
def test_grouped_conv3x3(H, D):
    L, N = _lib.lib(), 80
    torch.manual_seed(D)
    x = torch.randn(N, H, H, D, device=DEV).bfloat16()
    w = (torch.randn(D, D // 4, 3, 3, device=DEV) / (9 * D // 4) ** 0.5).bfloat16()
    gy = torch.randn(N, H, H, D, device=DEV).bfloat16()
    zeros = torch.zeros(64, device=DEV).bfloat16()
    xf, wf = x.float().permute(0, 3, 1, 2).requires_grad_(True), w.float().requires_grad_(True)
    yr = F.conv2d(xf, wf, None, 1, 1, 1, 4)
    yr.backward(gy.float().permute(0, 3, 1, 2))
    y = torch.full((N, H, H, D), float("nan"), device=DEV).bfloat16()
    assert L.cot_study_conv3x3g_nhwc(P(x), P(w.permute(0, 2, 3, 1).contiguous()), P(zeros), P(y), 0, N, H, H, D, D, 4, S()) == 0
    assert near(y, yr.detach().permute(0, 2, 3, 1))
    wt = w.view(4, D // 4, D // 4, 3, 3).flip(3, 4).permute(0, 2, 3, 4, 1).reshape(D, 3, 3, D // 4).contiguous()
    gx = torch.full((N, H, H, D), float("nan"), device=DEV).bfloat16()
    assert L.cot_study_conv3x3g_nhwc(P(gy), P(wt), P(zeros), P(gx), 0, N, H, H, D, D, 4, S()) == 0
    assert near(gx, xf.grad.permute(0, 2, 3, 1))
    L.cot_study_conv3x3g_nhwc_wgrad_workspace.restype = ctypes.c_size_t
    ws = torch.empty(L.cot_study_conv3x3g_nhwc_wgrad_workspace(N, H, H, D, D, 4, 0), device=DEV, dtype=torch.uint8)
    dwr = torch.full((D, 9, D // 4), float("nan"), device=DEV).bfloat16()
    assert L.cot_study_conv3x3g_nhwc_wgrad(P(x), P(gy), P(zeros), P(dwr), P(ws), N, H, H, D, D, 4, 0, S()) == 0
    assert near(dwr, wf.grad.permute(0, 2, 3, 1).reshape(D, 9, D // 4), 1e-2)


@pytest.mark.parametrize("HW,C,act,use_res", [(196, 256, 1, False), (196, 1024, 1, True), (49, 2048, 1, True), (196, 256, 2, False), (49, 512, 0, False)])
def test_batchnorm(HW, C, act, use_res):
    L, M, f32 = _lib.lib(), 80 * HW, ctypes.c_float
    torch.manual_seed(C + act)
    x = (torch.randn(M, C, device=DEV) * 1.5 + 0.7).bfloat16()
    res = torch.randn(M, C, device=DEV).bfloat16() if use_res else None
    dy = torch.randn(M, C, device=DEV).bfloat16()
    gamma, beta = torch.rand(C, device=DEV) + 0.5, torch.randn(C, device=DEV) * 0.2
    xr = x.float().requires_grad_(True)
    gr, br = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    z = F.batch_norm(xr.t().reshape(1, C, M, 1), None, None, gr, br, True, 0.1, 1e-5).reshape(C, M).t()
    if use_res:
        z = z + res.float()
    yr = {0: lambda t: t, 1: torch.relu, 2: F.silu}[act](z)
    y, mean, rstd = torch.full_like(x, float("nan")), torch.empty(C, device=DEV), torch.empty(C, device=DEV)
    ws = torch.empty(L.cot_study_bn_nhwc_workspace(M, C, BF), device=DEV)
    assert L.cot_study_bn_nhwc_forward(P(x), P(res), P(y), P(gamma), P(beta), P(mean), P(rstd), P(None), P(None), P(None), P(ws), M, C, f32(1e-5),
                                       f32(0.1), act, BF, S()) == 0
    assert ((y.float() - yr.detach()).abs() <= 2e-2 * (1 + yr.detach().abs())).all()
    # gradients with the kernel's own ReLU mask (the rounded output decides at |z| ~ 1e-3)
    g = dy.float() * ((y.float() > 0) if act == 1 else 1)
    if act == 2:
        g = torch.autograd.grad(F.silu(z), z, dy.float(), retain_graph=True)[0]
    z.backward(g)
    dx, dres = torch.full_like(x, float("nan")), (torch.full_like(x, float("nan")) if use_res else None)
    dg, db = torch.empty(C, device=DEV), torch.empty(C, device=DEV)
    assert L.cot_study_bn_nhwc_backward(P(dy), P(x), P(y), P(dx), P(dres), P(gamma), P(beta), P(mean), P(rstd), P(dg), P(db), P(ws), M, C, act, BF,
                                        S()) == 0
    assert near(dx, xr.grad, 3e-2) and torch.allclose(dg, gr.grad, rtol=2e-2, atol=2e-2 * gr.grad.abs().max().item())
    assert torch.allclose(db, br.grad, rtol=2e-2, atol=2e-2 * br.grad.abs().max().item())


@pytest.mark.parametrize("HW,D", [(196, 256), (49, 512)])
def test_group_norm9_radix_tail_and_layout(HW, D):
    L, N, G, f32 = _lib.lib(), 80, D // 8, ctypes.c_float
    torch.manual_seed(HW)
    C = 9 * G
    x, dy = (torch.randn(N, HW, C, device=DEV) * 1.7 + 0.4).bfloat16(), torch.randn(N, HW, C, device=DEV).bfloat16()
    gamma, beta = (1 + 0.3 * torch.randn(C, device=DEV)).bfloat16(), (0.2 * torch.randn(C, device=DEV)).bfloat16()
    xr, gr, br = x.float().requires_grad_(True), gamma.float().requires_grad_(True), beta.float().requires_grad_(True)
    yr = F.group_norm(xr.permute(0, 2, 1).reshape(N, C, HW, 1), G, gr, br, 1e-5).reshape(N, C, HW).permute(0, 2, 1)
    yr.backward(dy.float())
    y, mean, rstd = torch.full_like(x, float("nan")), torch.empty(N * G, device=DEV), torch.empty(N * G, device=DEV)
    assert L.cot_study_group_norm9_nhwc_forward(P(x), P(gamma), P(beta), P(y), P(mean), P(rstd), N, C, HW, f32(1e-5), BF, S()) == 0
    assert ((y.float() - yr.detach()).abs() <= 2e-2 * (1 + yr.detach().abs())).all()
    dx, dg, db = torch.full_like(x, float("nan")), torch.full_like(gamma, float("nan")), torch.full_like(beta, float("nan"))
    ws = torch.empty(N * C * 2, device=DEV)
    assert L.cot_study_group_norm9_nhwc_backward(P(dy), P(x), P(mean), P(rstd), P(gamma), P(dx), P(dg), P(db), P(ws), N, C, HW, BF, S()) == 0
    assert near(dx, xr.grad, 3e-2) and near(dg, gr.grad, 3e-2) and near(db, br.grad, 3e-2)
    # radix tail
    yy, kk, go = (torch.randn(N, HW, D, device=DEV).bfloat16() for _ in range(3))
    attn = torch.softmax(torch.randn(N, D, 2, device=DEV), 2).bfloat16()
    gap, out = torch.full((N, D), float("nan"), device=DEV).bfloat16(), torch.full_like(yy, float("nan"))
    assert L.cot_study_radix_nhwc_gap(P(yy), P(kk), P(gap), N, HW, D, BF, S()) == 0
    assert L.cot_study_radix_nhwc_mix(P(yy), P(kk), P(attn), P(out), N, HW, D, BF, S()) == 0
    assert near(gap, (yy.float() + kk.float()).mean(1)) and near(out, yy.float() * attn.float()[:, None, :, 0] + kk.float() * attn.float()[:, None, :, 1])
    ga, gy, gk = torch.full_like(attn, float("nan")), torch.full_like(yy, float("nan")), torch.full_like(yy, float("nan"))
    assert L.cot_study_radix_nhwc_mix_backward_reduce(P(go), P(yy), P(kk), P(ga), N, HW, D, BF, S()) == 0
    assert L.cot_study_radix_nhwc_mix_backward_apply(P(go), P(attn), P(gap), P(gy), P(gk), N, HW, D, BF, S()) == 0
    assert near(ga, torch.stack([(go.float() * yy.float()).sum(1), (go.float() * kk.float()).sum(1)], 2), 3e-2)
    assert near(gy, go.float() * attn.float()[:, None, :, 0] + gap.float()[:, None, :] / HW)
    # layout change
    t = torch.randn(N, D, HW, device=DEV).bfloat16()
    u, v = torch.full((N, HW, D), float("nan"), device=DEV).bfloat16(), torch.full((N, D, HW), float("nan"), device=DEV).bfloat16()
    assert L.cot_study_nchw_to_nhwc(P(t), P(u), N, D, HW, S()) == 0 and L.cot_study_nhwc_to_nchw(P(u), P(v), N, D, HW, S()) == 0
    assert torch.equal(u, t.permute(0, 2, 1).contiguous()) and torch.equal(v, t)

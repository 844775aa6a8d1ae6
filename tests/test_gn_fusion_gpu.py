"""GroupNorm-9 fused into its producer and its consumer on the GPU, at the benchmark batch (SURVEY 7.6 / VERDICT r3 J1): the
statistics come out of embed[3]'s epilogue (cot_conv1x1_forward_gn9 + cot_gn9_stats_finalize), the normalisation is applied in
the aggregation kernels' prologue (cot_agg_gn9_forward / _backward).  Checked against the unfused composition -- the same
convolution, csrc/group_norm9.hip, the plain aggregation kernels -- and against torch's GroupNorm statistics in fp32."""
import ctypes

import pytest
import torch

from cotnet_amd import _lib

pytestmark = pytest.mark.gpu
DEV = "cuda"
BF = _lib.COT_BF16


def P(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


@pytest.mark.parametrize("N,C,H", [(80, 64, 56), (80, 128, 28), (3, 64, 56)])
def test_fused_group_norm_matches_the_unfused_composition(N, C, H):
    L = _lib.lib()
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    torch.manual_seed(C + H + N)
    W, HW, Ch, Ce, G = H, H * H, C // 2, 9 * C // 8, C // 8
    e1 = torch.randn(N, Ch, H, W, device=DEV).bfloat16()
    w3, b3 = (torch.randn(Ce, Ch, device=DEV) / Ch ** 0.5).bfloat16(), torch.randn(Ce, device=DEV).bfloat16()
    gamma, beta = (1 + 0.3 * torch.randn(Ce, device=DEV)).bfloat16(), (0.2 * torch.randn(Ce, device=DEV)).bfloat16()
    assert L.cot_gn9_fused_covers(Ch, Ch, 0, HW, W) == 1
    part = torch.full((int(L.cot_gn9_stats_floats(N, Ce, HW)),), float("nan"), device=DEV)
    e3, e3b = (torch.full((N, Ce, H, W), float("nan"), device=DEV).bfloat16() for _ in range(2))
    assert L.cot_conv1x1_forward_gn9(P(e1), None, Ch, P(w3), P(b3), P(e3), P(part), N, Ch, Ce, HW, BF, st) == 0, L.cot_last_error()
    assert L.cot_conv1x1_forward(P(e1), None, Ch, P(w3), P(b3), P(e3b), N, Ch, Ce, HW, BF, st) == 0
    torch.cuda.synchronize()
    assert torch.equal(e3, e3b) and torch.isfinite(part).all()
    mean, rstd = torch.empty(N * G, device=DEV), torch.empty(N * G, device=DEV)
    assert L.cot_gn9_stats_finalize(P(part), P(mean), P(rstd), N, Ce, HW, 1e-5, st) == 0
    xg = e3.float().view(N, G, -1)
    assert torch.allclose(mean, xg.mean(2).flatten(), atol=2e-5, rtol=1e-5)
    assert torch.allclose(rstd, (xg.var(2, unbiased=False) + 1e-5).rsqrt().flatten(), atol=1e-5, rtol=2e-5)
    wn, m2, r2 = torch.empty_like(e3), torch.empty(N * G, device=DEV), torch.empty(N * G, device=DEV)
    assert L.cot_group_norm9_forward(P(e3), P(gamma), P(beta), P(wn), P(m2), P(r2), N, Ce, HW, 1e-5, BF, st) == 0
    v, go = torch.randn(N, C, H, W, device=DEV).bfloat16(), torch.randn(N, C, H, W, device=DEV).bfloat16()
    geo = _lib.AggGeom(N, C, H, W, 1, G, 3, 3, 1, 1, 1, 1, 1, 1)
    a1, a2 = torch.empty_like(v), torch.empty_like(v)
    # the same statistics on both sides: the prologue's arithmetic is the GroupNorm kernel's -> bit for bit
    assert L.cot_agg_gn9_forward(P(v), P(e3), P(m2), P(r2), P(gamma), P(beta), G, P(a1), ctypes.byref(geo), BF, st) == 0, L.cot_last_error()
    assert _lib.last_kernel() == "agg_fwd_nchw_k3_lds<gn9>"
    assert L.cot_agg_forward(P(v), P(wn), P(a2), ctypes.byref(geo), BF, 0, st) == 0
    gx1, gw1, gx2, gw2 = torch.empty_like(v), torch.empty_like(e3), torch.empty_like(v), torch.empty_like(e3)
    assert L.cot_agg_gn9_backward(P(go), P(v), P(e3), P(m2), P(r2), P(gamma), P(beta), G, P(gx1), P(gw1), ctypes.byref(geo), BF, st) == 0
    assert _lib.last_kernel() == "agg_bwd_nchw_k3_dot2<gx,gw,gn9>"
    assert L.cot_agg_backward(P(go), P(v), P(wn), P(gx2), P(gw2), ctypes.byref(geo), BF, 0, st) == 0
    torch.cuda.synchronize()
    assert torch.equal(a1, a2) and torch.equal(gx1, gx2) and torch.equal(gw1, gw2)
    # end to end with the epilogue statistics: the weights move by at most an ulp where mean / rstd differ in their last bits
    assert L.cot_agg_gn9_forward(P(v), P(e3), P(mean), P(rstd), P(gamma), P(beta), G, P(a1), ctypes.byref(geo), BF, st) == 0
    torch.cuda.synchronize()
    d = (a1.float() - a2.float()).abs()
    assert d.max() <= 2.0 ** -6 * a2.float().abs().max() and (d > 0).float().mean() < 0.01

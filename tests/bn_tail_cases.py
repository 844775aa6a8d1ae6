"""Shared case of the BatchNorm + SiLU -> radix-tail fusion tests (cot_radix_*_bn): run by tests/test_kernels_emulated.py through the
host-emulated library on CPU tensors and by tests/test_bn_tail_gpu.py through the product library on the MI355X."""
import ctypes

import torch

from cotnet_amd import _lib


def P(t):
    return ctypes.c_void_p(t.data_ptr())


def bn_tail_case(L, N, C, H, W, dtype, lay_k, sums=False, seed=21, report=None):
    """cot_bn_batch_stats + cot_radix_*_bn (BatchNorm + SiLU folded into the radix tail, models/cotnet.py:89-104) through library `L`
    on CPU tensors: forward against the unfused composition (cot_bn_act_forward, then cot_radix_gap_t / _mix_logits) -- bit for bit when
    the unfused call ran the streaming kernels, whose statistics the fused prologue repeats -- and both directions against torch autograd
    of the reference formula on the same rounded operands.  lay_k: k and the mix output / gk channel-major (the deep stages' blocks).
    Shared with tests/test_bn_tail_gpu.py (device tensors, the product library)."""
    torch.manual_seed(seed)
    dev = getattr(L, "_test_device", "cpu")
    HW = H * W
    dt = _lib.dtype_code(dtype)
    lp = dtype != torch.float32
    a = (torch.randn(N, C, H, W) * 1.3 + 0.4).to(dtype).to(dev)
    k, g = (torch.randn(N, C, H, W).to(dtype).to(dev) for _ in range(2))
    gamma, beta = (1 + 0.3 * torch.randn(C)).to(dev), (0.2 * torch.randn(C)).to(dev)
    eps, mom = 1e-5, 0.1
    cm = lambda t: t.permute(1, 0, 2, 3).contiguous()          # noqa: E731  dense [C][N][H][W] buffer
    uncm = lambda b: b.permute(1, 0, 2, 3).contiguous()        # noqa: E731
    kb = cm(k) if lay_k else k
    f32 = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)  # noqa: E731

    # unfused: BatchNorm + SiLU materialised, then the tail
    nws = int(L.cot_bn_act_workspace(N, C))
    mean0, rstd0, ws0 = f32(C), f32(C), f32(max(nws, 1))
    rm0, rv0, nbt0 = torch.zeros(C, device=dev), torch.ones(C, device=dev), torch.zeros((), dtype=torch.int64, device=dev)
    y = torch.empty_like(a)
    assert L.cot_bn_act_forward(P(a), None, P(y), P(gamma), P(beta), P(mean0), P(rstd0), P(rm0), P(rv0), P(nbt0), P(ws0), N, C, HW,
                                eps, mom, 2, dt, None) == 0, L.cot_last_error()
    gap0 = torch.empty(C, N, dtype=dtype, device=dev)
    assert L.cot_radix_gap_t_lay(P(y), P(kb), P(gap0), N, C, HW, 2 if lay_k else 0, dt, None) == 0
    logitsT = torch.randn(2 * C, N).to(dtype).to(dev)
    out0, attn0 = torch.empty_like(a), torch.empty(N, C, 2, dtype=dtype, device=dev)
    assert L.cot_radix_mix_logits_lay(P(y), P(kb), P(logitsT), P(out0), P(attn0), N, C, HW, 6 if lay_k else 0, dt, None) == 0

    # fused: the statistics alone (either route), y formed on load
    mean, rstd, ws = f32(C), f32(C), f32(max(nws, 1))
    rm, rv, nbt = torch.zeros(C, device=dev), torch.ones(C, device=dev), torch.zeros((), dtype=torch.int64, device=dev)
    gap = torch.empty(C, N, dtype=dtype, device=dev)
    if sums:  # chunk sums, finalized in the pooling kernel's prologue (the nodes' route)
        assert L.cot_bn_stats_sums(P(a), P(ws), N, C, HW, dt, None) == 0, L.cot_last_error()
        assert L.cot_radix_gap_t_bn(P(a), P(kb), P(gap), P(gamma), P(beta), P(mean), P(rstd), P(rm), P(rv), P(nbt), P(ws), N, C, HW, eps,
                                    mom, 2 if lay_k else 0, dt, None) == 0, L.cot_last_error()
    else:     # statistics finalized by their own launches, read by the pooling kernel
        assert L.cot_bn_batch_stats(P(a), P(mean), P(rstd), P(rm), P(rv), P(nbt), P(ws), N, C, HW, eps, mom, dt, None) == 0, L.cot_last_error()
        assert L.cot_radix_gap_t_bn(P(a), P(kb), P(gap), P(gamma), P(beta), P(mean), P(rstd), None, None, None, None, N, C, HW, eps, mom,
                                    2 if lay_k else 0, dt, None) == 0, L.cot_last_error()
    out, attn = torch.empty_like(a), torch.empty(N, C, 2, dtype=dtype, device=dev)
    assert L.cot_radix_mix_logits_bn(P(a), P(kb), P(logitsT), P(out), P(attn), P(gamma), P(beta), P(mean), P(rstd), N, C, HW,
                                     6 if lay_k else 0, dt, None) == 0, L.cot_last_error()
    af = a.float().cpu()
    assert torch.allclose(mean.cpu(), af.mean((0, 2, 3)), atol=1e-5, rtol=1e-5)
    assert torch.allclose(rstd.cpu(), 1 / torch.sqrt(af.var((0, 2, 3), unbiased=False) + eps), atol=1e-4, rtol=1e-4)
    assert int(nbt) == 1 and torch.allclose(rm.cpu(), rm0.cpu(), atol=1e-6) and torch.allclose(rv.cpu(), rv0.cpu(), atol=1e-5, rtol=1e-5)
    same_stats = torch.equal(mean, mean0) and torch.equal(rstd, rstd0)
    if same_stats and lp:  # (the unfused call took the streaming kernels: the same chunk statistics, merged in the same order.  bf16 storage:
        # y is rounded to bf16 on both routes, which absorbs the last-place differences of the two kernels' fp32 expressions; fp32 storage
        # keeps them -- found by the fuzz on populations whose statistics happened to agree to the bit)
        assert torch.equal(gap, gap0) and torch.equal(out, out0) and torch.equal(attn, attn0)
    else:
        tol = 3e-2 if lp else 1e-4
        assert torch.allclose(gap.float(), gap0.float(), atol=tol, rtol=tol) and torch.allclose(out.float(), out0.float(), atol=tol, rtol=tol)

    # the reference formula in fp32 on the same operands (y NOT rounded here: tolerance below)
    a_r, k_r = af.clone().requires_grad_(True), k.float().cpu().requires_grad_(True)
    gam_r, bet_r = gamma.cpu().clone().requires_grad_(True), beta.cpu().clone().requires_grad_(True)
    z = torch.nn.functional.batch_norm(a_r, None, None, gam_r, bet_r, True, 0.0, eps)
    z.retain_grad()
    y_r = torch.nn.functional.silu(z)
    gap_r = (y_r + k_r).mean((2, 3))                                          # [N, C]
    lg = logitsT.float().cpu().t().reshape(N, C, 2).requires_grad_(True)
    at = torch.softmax(lg, dim=2)
    out_r = y_r * at[:, :, 0, None, None] + k_r * at[:, :, 1, None, None]
    tol = 3e-2 if lp else 2e-4
    assert torch.allclose(gap.float().cpu(), gap_r.detach().t(), atol=tol, rtol=tol)
    out_n = uncm(out.view(C, N, H, W)) if lay_k else out  # (written channel-major into a buffer of a's size)
    assert torch.allclose(out_n.float().cpu(), out_r.detach(), atol=tol, rtol=tol)
    ggapT = (0.5 * torch.randn(C, N)).to(dtype).to(dev)
    gb = cm(g) if lay_k else g
    (out_r * g.float().cpu()).sum().backward(retain_graph=True)
    want_gl = lg.grad.reshape(N, 2 * C).t().clone()
    (gap_r * ggapT.float().cpu().t()).sum().backward()

    glogT, tsum = torch.empty(2 * C, N, dtype=dtype, device=dev), f32(N * C * 4)
    assert L.cot_radix_mix_backward_reduce_bn(P(gb), P(a), P(kb), P(attn), P(glogT), P(tsum), P(gamma), P(beta), P(mean), P(rstd), N, C,
                                              HW, (1 if lay_k else 0) | (4 if lay_k else 0), dt, None) == 0, L.cot_last_error()
    assert (glogT.float().cpu() - want_gl).abs().max() <= 3 * tol * (1 + want_gl.abs().max())
    ga, gk = torch.full_like(a, float("nan")), torch.full_like(kb, float("nan"))
    dgam, dbet = f32(C), f32(C)
    assert L.cot_radix_mix_backward_apply_bn(P(gb), P(a), P(attn), P(ggapT), P(tsum), P(ga), P(gk), P(gamma), P(beta), P(mean), P(rstd),
                                             P(dgam), P(dbet), N, C, HW, (1 if lay_k else 0) | (4 if lay_k else 0), dt,
                                             None) == 0, L.cot_last_error()
    gk_n = uncm(gk) if lay_k else gk
    scale = lambda t: float(t.abs().max()) + 1e-6  # noqa: E731
    btol = 4e-2 if lp else 1e-3
    assert (gk_n.float().cpu() - k_r.grad).abs().max() <= btol * scale(k_r.grad)
    assert (ga.float().cpu() - a_r.grad).abs().max() <= btol * scale(a_r.grad), (ga.float().cpu() - a_r.grad).abs().max() / scale(a_r.grad)
    # dgamma / dbeta are sums of N * HW signed terms dz (* z_hat): a sum that cancels to almost nothing (one channel, small populations) still
    # carries the rounding of its terms -- the yardstick is the sum's natural size rms(dz) * sqrt(count) next to the largest gradient
    noise = float(z.grad.pow(2).mean().sqrt()) * (N * HW) ** 0.5
    if report is not None:
        report.append(((dgam.cpu() - gam_r.grad).abs().max().item(), (dbet.cpu() - bet_r.grad).abs().max().item(), scale(gam_r.grad),
                       scale(bet_r.grad), noise))
        return same_stats
    assert (dgam.cpu() - gam_r.grad).abs().max() <= btol * (scale(gam_r.grad) + noise)
    assert (dbet.cpu() - bet_r.grad).abs().max() <= btol * (scale(bet_r.grad) + noise)
    return same_stats


def rowstats_case(L, N, C, H, W, gn, seed=5):
    """cot_agg_forward_rowstats (the aggregation forward that also emits the following BatchNorm's per-row sums) + cot_bn_rowstats_finalize
    against the plain entry points: the output bit for bit, the row sums against torch on the stored values, mean / rstd / running
    statistics against torch's batch statistics.  gn: GroupNorm-9 prologue (w = raw logits) as in cot_agg_gn9_forward."""
    torch.manual_seed(seed)
    dev = getattr(L, "_test_device", "cpu")
    wC, HW = C // 8, H * W
    dt = _lib.COT_BF16
    x = torch.randn(N, C, H, W).bfloat16().to(dev)
    w = (0.4 * torch.randn(N, 1, wC, 9, H, W) + 0.1).bfloat16().to(dev)
    geom = _lib.AggGeom(N, C, H, W, 1, wC, 3, 3, 1, 1, 1, 1, 1, 1)
    out0, out = torch.empty_like(x), torch.full_like(x, float("nan"))
    rows = torch.full((int(L.cot_agg_rowstats_floats(N, C, H)),), float("nan"), dtype=torch.float32, device=dev)
    if gn:
        G = wC  # one GroupNorm group per weight channel: 9 taps each
        gm, gr = torch.randn(N * G).to(dev) * 0.1, (1 + 0.2 * torch.rand(N * G)).to(dev)
        gam, bet = (1 + 0.3 * torch.randn(9 * G)).bfloat16().to(dev), (0.2 * torch.randn(9 * G)).bfloat16().to(dev)
        assert L.cot_agg_gn9_forward(P(x), P(w), P(gm), P(gr), P(gam), P(bet), G, P(out0), ctypes.byref(geom), dt, None) == 0, L.cot_last_error()
        rc = L.cot_agg_forward_rowstats(P(x), P(w), P(out), P(rows), P(gm), P(gr), P(gam), P(bet), G, ctypes.byref(geom), dt, None)
    else:
        assert L.cot_agg_forward(P(x), P(w), P(out0), ctypes.byref(geom), dt, _lib.COT_NCHW, None) == 0, L.cot_last_error()
        rc = L.cot_agg_forward_rowstats(P(x), P(w), P(out), P(rows), None, None, None, None, 0, ctypes.byref(geom), dt, None)
    assert rc == 0, L.cot_last_error()
    assert L.cot_last_kernel().decode().endswith("rowstats>")
    assert torch.equal(out, out0)
    of = out.float().cpu()
    r = rows.cpu().view(N, C, H, 2)
    assert torch.allclose(r[..., 0], of.sum(3), atol=1e-4, rtol=1e-5) and torch.allclose(r[..., 1], (of * of).sum(3), atol=1e-4, rtol=1e-5)
    f32 = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)  # noqa: E731
    mean, rstd = f32(C), f32(C)
    rm, rv, nbt = torch.zeros(C, device=dev), torch.ones(C, device=dev), torch.zeros((), dtype=torch.int64, device=dev)
    assert L.cot_bn_rowstats_finalize(P(rows), P(mean), P(rstd), P(rm), P(rv), P(nbt), N, C, H, W, 1e-5, 0.1, None) == 0, L.cot_last_error()
    var = of.var((0, 2, 3), unbiased=False)
    assert torch.allclose(mean.cpu(), of.mean((0, 2, 3)), atol=2e-6, rtol=1e-5)
    assert torch.allclose(rstd.cpu(), 1 / torch.sqrt(var + 1e-5), atol=1e-5, rtol=1e-5)
    cnt = N * HW
    assert int(nbt) == 1 and torch.allclose(rm.cpu(), 0.1 * of.mean((0, 2, 3)), atol=1e-6, rtol=1e-5)
    assert torch.allclose(rv.cpu(), 0.9 + 0.1 * var * cnt / max(cnt - 1, 1), atol=1e-5, rtol=1e-5)


def relu_res_case(L, N, Ci, Co, HW, seed=7):
    """cot_conv1x1_backward_data_relu_res (conv1's data gradient + the residual's gradient gout * [y > 0] from bn3's sign mask) against
    the two-step form it replaces -- the masked gradient materialised, then cot_conv1x1_backward_data(accumulate = 1): bit for bit."""
    torch.manual_seed(seed)
    dev = getattr(L, "_test_device", "cpu")
    dt = _lib.COT_BF16
    assert L.cot_conv1x1_backward_data_relu_res_covers(N, Ci, Co, HW, dt) == 1
    gy = torch.randn(N, Co, HW).bfloat16().to(dev)
    w = (torch.randn(Co, Ci) / Co ** 0.5).bfloat16().to(dev)
    gout, y = torch.randn(N, Ci, HW).bfloat16().to(dev), torch.randn(N, Ci, HW).to(dev)
    bits = (y.reshape(-1, 8) > 0).to(torch.int32)
    mask = (bits * (2 ** torch.arange(8, device=dev, dtype=torch.int32))).sum(1).to(torch.uint8).contiguous()
    assert mask.numel() == int(L.cot_bn_relu_mask_bytes(N, Ci, HW, dt))
    gx0 = (gout.float() * (y > 0)).bfloat16().contiguous()
    ws = torch.empty(max(int(L.cot_conv1x1_workspace(N, Ci, Co, HW, 0)), 16), dtype=torch.uint8, device=dev)
    assert L.cot_conv1x1_backward_data(P(gy), P(w), P(gx0), None, Ci, 1, P(ws), N, Ci, Co, HW, dt, None) == 0, L.cot_last_error()
    gx = torch.full_like(gout, float("nan"))
    assert L.cot_conv1x1_backward_data_relu_res(P(gy), P(w), P(gx), P(gout), P(mask), N, Ci, Co, HW, dt, None) == 0, L.cot_last_error()
    assert torch.equal(gx, gx0), (gx.float() - gx0.float()).abs().max().item()
    ref = torch.einsum("oc,nop->ncp", w.float().cpu(), gy.float().cpu()) + gout.float().cpu() * (y.cpu() > 0)
    assert (gx.float().cpu() - ref).abs().max() <= 2e-2 * ref.abs().max()

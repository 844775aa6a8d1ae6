"""Cases for the per-tensor-layout entry points (cot_*_lay, include/cotnet_amd.h), shared by the host-emulated run
(tests/test_layouts_emulated.py) and the GPU run (tests/test_layouts_gpu.py).  Every case computes the same function twice: on
NCHW tensors through the ordinary entry point, and with some tensors channel-major ([C][N][HW]) through the _lay entry point; the
arithmetic and the order of every sum are the same, so the results must be BIT-identical after undoing the permutation.
(models/cotnet.py:43-62, :89-104 are the module sequences these kernels serve.)"""
import ctypes
import itertools

import torch

from cotnet_amd import _lib

BF = _lib.COT_BF16


def P(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def cm(t):
    """[N, C, ...] -> channel-major buffer [C, N, ...] (dense)"""
    return t.transpose(0, 1).contiguous()


def uncm(t):
    return t.transpose(0, 1).contiguous()


_KEEP = []  # the permuted copies handed to a kernel must outlive the call (P() takes a raw pointer)


def put(t, bit):
    r = cm(t) if bit else t.contiguous()
    _KEEP.append(r)
    if len(_KEEP) > 64:
        _sync(r.device)
        del _KEEP[:32]
    return r


def get(buf, bit):
    return uncm(buf) if bit else buf


def _ok(L, rc):
    assert rc == 0, L.cot_last_error().decode()


def bn_forward_case(L, dev, st, N, C, HW, act, with_res, with_y2, with_ps, seed=0):
    g = torch.Generator().manual_seed(seed + N + C + HW)
    x = (torch.randn(N, C, HW, generator=g) * 1.5 + 0.3).bfloat16().to(dev)
    res = torch.randn(N, C, HW, generator=g).bfloat16().to(dev) if with_res else None
    gamma, beta = (torch.rand(C, generator=g) + 0.5).to(dev), (torch.randn(C, generator=g) * 0.2).to(dev)
    ps = ((torch.rand(N, generator=g) < 0.7).float() / 0.7).to(dev) if with_ps else None
    assert L.cot_bn_act_lay_covers(N, C, HW, BF) == 1

    def stats():
        return (torch.empty(C, device=dev), torch.empty(C, device=dev), torch.zeros(C, device=dev), torch.ones(C, device=dev),
                torch.zeros((), dtype=torch.int64, device=dev))
    mean, rstd, rm, rv, nbt = stats()
    ws = torch.empty(max(1, int(L.cot_bn_act_workspace(N, C))), device=dev)
    y = torch.full((N, C, HW), float("nan")).bfloat16().to(dev)
    _ok(L, L.cot_bn_act_forward_ps(P(x), P(res), P(y), P(gamma), P(beta), P(mean), P(rstd), P(rm), P(rv), P(nbt), P(ws), P(ps), N, C, HW,
                                   1e-5, 0.1, act, BF, st))
    for bits in itertools.product((0, 1), repeat=4):
        bx, br, by, by2 = bits
        if (not with_res and br) or (not with_y2 and by2):
            continue
        lay = bx | (br << 1) | (by << 2) | (by2 << 3)
        m2, r2, rm2, rv2, nbt2 = stats()
        yl = torch.full_like(y, float("nan"))
        y2l = torch.full_like(y, float("nan")) if with_y2 else None
        _ok(L, L.cot_bn_act_forward_lay(P(put(x, bx)), P(put(res, br)) if with_res else None, P(yl), P(y2l), P(gamma), P(beta), P(m2), P(r2),
                                        P(rm2), P(rv2), P(nbt2), P(ps), N, C, HW, 1e-5, 0.1, act, lay, BF, st))
        _sync(dev)
        assert torch.equal(get(yl.view(C, N, HW) if by else yl, by), y), ("y", bits)
        if with_y2:
            assert torch.equal(get(y2l.view(C, N, HW) if by2 else y2l, by2), y), ("y2", bits)
        assert torch.equal(m2, mean) and torch.equal(r2, rstd) and torch.equal(rm2, rm) and torch.equal(rv2, rv) and int(nbt2) == 1
    return x, res, y, gamma, beta, mean, rstd, ps


def bn_backward_case(L, dev, st, N, C, HW, act, with_res, with_dy2, with_ps, seed=0):
    x, res, y, gamma, beta, mean, rstd, ps = bn_forward_case(L, dev, st, N, C, HW, act, with_res, False, with_ps, seed)
    g = torch.Generator().manual_seed(seed + 17)
    dy = torch.randn(N, C, HW, generator=g).bfloat16().to(dev)
    dy2 = torch.randn(N, C, HW, generator=g).bfloat16().to(dev) if with_dy2 else None
    dysum = (dy.float() + dy2.float()).bfloat16() if with_dy2 else dy  # (the kernel: fp32 sum, one rounding)
    ws = torch.empty(max(1, int(L.cot_bn_act_workspace(N, C))), device=dev)
    dx = torch.full_like(x, float("nan"))
    dres = torch.full_like(x, float("nan")) if with_res else None
    dg, db = torch.empty(C, device=dev), torch.empty(C, device=dev)
    ysave = y if (with_res and act == 1) else None
    _ok(L, L.cot_bn_act_backward_ps(P(dysum), P(x), P(ysave), P(dx), P(dres), P(gamma), P(beta), P(mean), P(rstd), P(dg), P(db), P(ws),
                                    P(ps), N, C, HW, act, BF, st))
    for bits in itertools.product((0, 1), repeat=6):
        bdy, bdy2, bx, by, bdx, bdr = bits
        if (not with_dy2 and bdy2) or (ysave is None and by) or (not with_res and bdr):
            continue
        lay = bdy | (bdy2 << 1) | (bx << 2) | (by << 3) | (bdx << 4) | (bdr << 5)
        dxl = torch.full_like(x, float("nan"))
        drl = torch.full_like(x, float("nan")) if with_res else None
        dg2, db2 = torch.empty(C, device=dev), torch.empty(C, device=dev)
        _ok(L, L.cot_bn_act_backward_lay(P(put(dy, bdy)), P(put(dy2, bdy2)) if with_dy2 else None, P(put(x, bx)),
                                         P(put(ysave, by)) if ysave is not None else None, P(dxl), P(drl), P(gamma), P(beta), P(mean),
                                         P(rstd), P(dg2), P(db2), P(ps), N, C, HW, act, lay, BF, st))
        _sync(dev)
        assert torch.equal(get(dxl.view(C, N, HW) if bdx else dxl, bdx), dx), ("dx", bits)
        if with_res:
            assert torch.equal(get(drl.view(C, N, HW) if bdr else drl, bdr), dres), ("dres", bits)
        assert torch.equal(dg2, dg) and torch.equal(db2, db)


def radix_case(L, dev, st, N, C, HW, seed=0):
    g = torch.Generator().manual_seed(seed + N + C + HW)
    rnd = lambda *s: torch.randn(*s, generator=g).bfloat16().to(dev)  # noqa: E731
    y, k, gout = rnd(N, C, HW), rnd(N, C, HW), rnd(N, C, HW)
    logitsT, ggapT = rnd(2 * C, N), rnd(C, N)
    gapT = torch.empty(C, N, dtype=torch.bfloat16, device=dev)
    _ok(L, L.cot_radix_gap_t(P(y), P(k), P(gapT), N, C, HW, BF, st))
    out, attn = torch.empty_like(y), torch.empty(N, C, 2, dtype=torch.bfloat16, device=dev)
    _ok(L, L.cot_radix_mix_logits(P(y), P(k), P(logitsT), P(out), P(attn), N, C, HW, BF, st))
    glog = torch.empty(2 * C, N, dtype=torch.bfloat16, device=dev)
    _ok(L, L.cot_radix_mix_backward_reduce(P(gout), P(y), P(k), P(attn), P(glog), N, C, HW, BF, st))
    gy, gk = torch.empty_like(y), torch.empty_like(y)
    _ok(L, L.cot_radix_mix_backward_apply(P(gout), P(attn), P(ggapT), P(gy), P(gk), N, C, HW, BF, st))
    for b0, b1, b2 in itertools.product((0, 1), repeat=3):
        lay = b0 | (b1 << 1) | (b2 << 2)
        nan = lambda t: torch.full_like(t, float("nan"))  # noqa: E731
        if not b2:
            g2 = nan(gapT)
            _ok(L, L.cot_radix_gap_t_lay(P(put(y, b0)), P(put(k, b1)), P(g2), N, C, HW, lay, BF, st))
            _sync(dev)
            assert torch.equal(g2, gapT), ("gap_t", lay)
            if not b1:
                h0, h1 = nan(gapT), nan(gapT)
                _ok(L, L.cot_radix_gap_t(P(y), None, P(h0), N, C, HW, BF, st))
                _ok(L, L.cot_radix_gap_t_lay(P(put(y, b0)), None, P(h1), N, C, HW, lay, BF, st))
                _sync(dev)
                assert torch.equal(h0, h1)
        o2, a2 = nan(out), nan(attn)
        _ok(L, L.cot_radix_mix_logits_lay(P(put(y, b0)), P(put(k, b1)), P(logitsT), P(o2), P(a2), N, C, HW, lay, BF, st))
        _sync(dev)
        assert torch.equal(get(o2.view(C, N, HW) if b2 else o2, b2), out) and torch.equal(a2, attn), ("mix_logits", lay)
        gl2 = nan(glog)
        _ok(L, L.cot_radix_mix_backward_reduce_lay(P(put(gout, b0)), P(put(y, b1)), P(put(k, b2)), P(attn), P(gl2), N, C, HW, lay, BF, st))
        _sync(dev)
        assert torch.equal(gl2, glog), ("bwd_reduce", lay)
        gy2, gk2 = nan(gy), nan(gk)
        _ok(L, L.cot_radix_mix_backward_apply_lay(P(put(gout, b0)), P(attn), P(ggapT), P(gy2), P(gk2), N, C, HW, lay, BF, st))
        _sync(dev)
        assert torch.equal(get(gy2.view(C, N, HW) if b1 else gy2, b1), gy), ("bwd_apply gy", lay)
        assert torch.equal(get(gk2.view(C, N, HW) if b2 else gk2, b2), gk), ("bwd_apply gk", lay)


def gn9_case(L, dev, st, N, G, HW, seed=0):
    C = 9 * G
    g = torch.Generator().manual_seed(seed + N + C + HW)
    rnd = lambda *s: torch.randn(*s, generator=g).bfloat16().to(dev)  # noqa: E731
    x, dy = rnd(N, C, HW), rnd(N, C, HW)
    gamma, beta = (1 + 0.3 * torch.randn(C, generator=g)).bfloat16().to(dev), (0.2 * torch.randn(C, generator=g)).bfloat16().to(dev)
    y = torch.full_like(x, float("nan"))
    mean, rstd = torch.empty(N * G, device=dev), torch.empty(N * G, device=dev)
    _ok(L, L.cot_group_norm9_forward(P(x), P(gamma), P(beta), P(y), P(mean), P(rstd), N, C, HW, 1e-5, BF, st))
    dx = torch.full_like(x, float("nan"))
    dg, db = torch.empty(C, dtype=torch.bfloat16, device=dev), torch.empty(C, dtype=torch.bfloat16, device=dev)
    ws = torch.empty(2 * N * C, device=dev)
    _ok(L, L.cot_group_norm9_backward(P(dy), P(x), P(mean), P(rstd), P(gamma), P(dx), P(dg), P(db), P(ws), N, C, HW, BF, st))
    for b0, b1, b2 in itertools.product((0, 1), repeat=3):
        if not b2:
            y2, m2, r2 = torch.full_like(x, float("nan")), torch.empty_like(mean), torch.empty_like(rstd)
            _ok(L, L.cot_group_norm9_forward_lay(P(put(x, b0)), P(gamma), P(beta), P(y2), P(m2), P(r2), N, C, HW, 1e-5, b0 | (b1 << 1), BF, st))
            _sync(dev)
            assert torch.equal(get(y2.view(C, N, HW) if b1 else y2, b1), y) and torch.equal(m2, mean) and torch.equal(r2, rstd), ("gn fwd", b0, b1)
        dx2 = torch.full_like(x, float("nan"))
        dg2, db2 = torch.empty_like(dg), torch.empty_like(db)
        _ok(L, L.cot_group_norm9_backward_lay(P(put(dy, b0)), P(put(x, b1)), P(mean), P(rstd), P(gamma), P(dx2), P(dg2), P(db2), P(ws), N, C,
                                              HW, b0 | (b1 << 1) | (b2 << 2), BF, st))
        _sync(dev)
        assert torch.equal(get(dx2.view(C, N, HW) if b2 else dx2, b2), dx) and torch.equal(dg2, dg) and torch.equal(db2, db), ("gn bwd", b0, b1, b2)


def _sync(dev):
    if str(dev).startswith("cuda"):
        torch.cuda.synchronize()


def _lib_last(L):
    return L.cot_last_kernel().decode()


def bn_epilogue_case(L, dev, st, N, Ci, Co, HW, act, res, mask):
    """conv1x1 -> BatchNorm with the statistics out of the convolution's epilogue (cot_conv1x1_forward_stats + cot_bn_tile_stats_finalize +
    cot_bn_act_apply_forward; models/cotnet.py:59-62, :248-262) against the separate pair (cot_conv1x1_forward + cot_bn_act_forward[_mask]):
    the same convolution output bit for bit, statistics equal to fp32 rounding (fp64 sums of the stored values vs the two-pass fp32
    form), outputs within one bf16 rounding, identical sign masks wherever the outputs agree"""
    g = torch.Generator().manual_seed(N + Ci + Co)
    x = torch.randn(N, Ci, HW, generator=g).bfloat16().to(dev)
    w = (torch.randn(Co, Ci, generator=g) / Ci ** 0.5).bfloat16().to(dev)
    r = torch.randn(N, Co, HW, generator=g).bfloat16().to(dev) if res else None
    gamma, beta = (torch.rand(Co, generator=g) + 0.5).to(dev), (torch.randn(Co, generator=g) * 0.2).to(dev)
    assert L.cot_conv1x1_stats_covers(Ci, Ci, 0, HW) == 1 and L.cot_conv1x1_stats_covers(Ci, Ci, 0, 196) == 0
    part = torch.full((int(L.cot_gn9_stats_floats(N, Co, HW)),), float("nan"), device=dev)
    y0, y1 = torch.full((N, Co, HW), float("nan"), device=dev).bfloat16(), torch.full((N, Co, HW), float("nan"), device=dev).bfloat16()
    assert L.cot_conv1x1_forward_stats(P(x), None, Ci, P(w), None, P(y0), P(part), N, Ci, Co, HW, BF, st) == 0, L.cot_last_error().decode()
    assert L.cot_conv1x1_forward(P(x), None, Ci, P(w), None, P(y1), N, Ci, Co, HW, BF, st) == 0
    _sync(dev)
    assert torch.equal(y0, y1) and torch.isfinite(part).all()
    mean, rstd, rm, rv = torch.empty(Co, device=dev), torch.empty(Co, device=dev), torch.zeros(Co, device=dev), torch.ones(Co, device=dev)
    nbt = torch.zeros((), dtype=torch.int64, device=dev)
    assert L.cot_bn_tile_stats_finalize(P(part), P(mean), P(rstd), P(rm), P(rv), P(nbt), N, Co, HW, 1e-5, 0.1, st) == 0
    z = torch.full_like(y0, float("nan"), device=dev)
    mk = torch.zeros(int(L.cot_bn_relu_mask_bytes(N, Co, HW, BF)), dtype=torch.uint8, device=dev) if mask else None
    assert L.cot_bn_act_apply_forward(P(y0), P(r) if res else None, P(z), P(mk) if mask else None, P(gamma), P(beta), P(mean), P(rstd), N, Co,
                                      HW, act, BF, st) == 0, L.cot_last_error().decode()
    m2, r2, rm2, rv2 = torch.empty(Co, device=dev), torch.empty(Co, device=dev), torch.zeros(Co, device=dev), torch.ones(Co, device=dev)
    nbt2 = torch.zeros((), dtype=torch.int64, device=dev)
    ws = torch.empty(max(1, int(L.cot_bn_act_workspace(N, Co))), device=dev)
    z2 = torch.full_like(y0, float("nan"), device=dev)
    mk2 = torch.zeros_like(mk) if mask else None
    if mask:
        assert L.cot_bn_act_forward_mask(P(y0), P(r), P(z2), P(mk2), P(gamma), P(beta), P(m2), P(r2), P(rm2), P(rv2), P(nbt2), P(ws), None, N,
                                         Co, HW, 1e-5, 0.1, act, BF, st) == 0, L.cot_last_error().decode()
    else:
        assert L.cot_bn_act_forward(P(y0), P(r) if res else None, P(z2), P(gamma), P(beta), P(m2), P(r2), P(rm2), P(rv2), P(nbt2), P(ws), N, Co,
                                    HW, 1e-5, 0.1, act, BF, st) == 0, L.cot_last_error().decode()
    _sync(dev)
    assert torch.allclose(mean, m2, atol=1e-6, rtol=1e-5) and torch.allclose(rstd, r2, atol=0, rtol=2e-5)
    assert torch.allclose(rm, rm2, atol=1e-6, rtol=1e-5) and torch.allclose(rv, rv2, atol=1e-6, rtol=2e-5) and int(nbt) == 1
    d = (z.float() - z2.float()).abs()
    # (one bf16 rounding of the output; where the normalised value and the residual cancel, the last-bit difference of the two
    # statistics paths shows in absolute terms: 1e-4 on values of order one)
    assert (d <= 2.0 ** -7 * z2.float().abs() + 1e-4).all() and (d > 0).float().mean() < 0.10
    if mask:
        same = (z == z2).view(-1, 8).all(1)
        assert torch.equal(mk[same], mk2[same])

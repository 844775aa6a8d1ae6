"""Element-wise parity of EVERY (kernel, shape) the headline dispatches, at the benchmark batch.

tests/test_dispatch_table.py pins WHICH kernel instance serves each layer of BASELINE config 2/3 (CoTNet-50 224^2, B = 80) --
in dry-run mode, no numbers.  The op-level GPU tests use N <= 8, where the dispatcher picks other template instances, split
counts, images per workgroup and tile orders.  Here every `cotnet50_b80_224` key of tests/golden/dispatch_table.json is

  1. re-issued against the real library in dry-run mode with the real pointers: the launches must be the pinned ones
     (same instance, grid and block as the table: what the numbers below are about is what the benchmark runs), then
  2. EXECUTED at that very shape through the C ABI and compared element by element with fp32 torch on the same bf16-rounded
     operands: |a - b| <= 1e-2 * (|b| + mean|b|) (the bar of tests/test_conv1x1_gpu.py: bf16 output rounding is 2^-9
     relative, the rest is summation order), BatchNorm / GroupNorm parameter gradients and statistics at 1e-3 relative;
     the aggregation against the oracle (oracle/agg_oracle.c, the reference's loop order) for all four stages -- bf16 inputs
     within one output rounding, integer-valued inputs bit-exact.

The test ids are the table's keys."""
import ctypes
import json
import os

import pytest
import torch
import torch.nn.functional as F

from cotnet_amd import _lib

pytestmark = pytest.mark.gpu
DEV = "cuda"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TABLE = json.load(open(os.path.join(ROOT, "tests", "golden", "dispatch_table.json")))
CFG = "cotnet50_b80_224"
KEYS = sorted(k for k in TABLE if k.startswith(CFG + " "))
# BASELINE configs 4 / 5 at their one-GPU benchmark batch (bench.py's `secondary` lines): the same walk.  (Entries without a pass
# suffix are the table's notes about modules outside the library -- strided 3x3 convolutions of a deep stem.)
OTHER_CFGS = ("cotnext101_2x48d_b64_224", "se_cotnetd_152_L_b64_320")
OTHER_KEYS = sorted(k for k in TABLE if k.split(" ")[0] in OTHER_CFGS and len(k.split(" ", 3)) == 4)
BF = _lib.COT_BF16


def P(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def _st():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _close(a, b, rel=1e-2):
    b = b.float()
    bad = (a.float() - b).abs() > rel * (b.abs() + b.abs().mean())
    assert not bad.any(), (int(bad.sum()), (a.float() - b).abs().max().item(), b.abs().mean().item())


def _rel(a, b, rel):
    b = b.float()
    e = (a.float() - b).abs().max().item()
    assert e <= rel * (b.abs().max().item() + 1e-6), (e, b.abs().max().item())


def _randn(*shape, seed=0, scale=1.0):
    g = torch.Generator(device=DEV).manual_seed(seed)
    return (torch.randn(*shape, device=DEV, generator=g) * scale).bfloat16()


def _gamma_beta(C, seed):
    """affine parameters from a generator of their own: the cases must not depend on which tests drew from the global generator before them
    (round 6: a subset run in another order put one bncm case's dbeta 0.7 % off -- ReLU decisions of near-zero pre-activations, see below)"""
    g = torch.Generator(device=DEV).manual_seed(7919 + seed)
    return (torch.rand(C, device=DEV, generator=g) + 0.5).float(), (torch.randn(C, device=DEV, generator=g) * 0.2).float()


def _param_grads_under_the_kernels_mask(x, dsum, y, eps=1e-5):
    """dgamma / dbeta of relu(bn(x)) in fp32 with the ReLU decisions the KERNEL took (the sign of its stored output): the fp32 reference
    decides a pre-activation below the inputs' bf16 resolution differently now and then, and each such element moves a channel's sum by
    |dy| -- per-element that is masked out of the dx comparison, in a sum it cannot be"""
    xf = x.float()
    mu = xf.mean((0, 2, 3), keepdim=True)
    xhat = (xf - mu) * torch.rsqrt(xf.var((0, 2, 3), unbiased=False, keepdim=True) + eps)
    dz = dsum * (y.float() > 0)
    return (dz * xhat).sum((0, 2, 3)), dz.sum((0, 2, 3))


class _Pinned:
    """issue(fn) runs `fn()` (one C-ABI call) twice: dry (launch log compared with the table's entry) and for real"""

    def __init__(self, key):
        self.key, self.L = key, _lib.lib()

    def issue(self, fn):
        from tests.test_dispatch_table import _short
        L = self.L
        buf = ctypes.create_string_buffer(1 << 14)
        L.cot_launch_log(buf, len(buf))  # (drain)
        assert L.cot_set_tuning(26, 1) == 0
        try:
            rc = fn()
            L.cot_launch_log(buf, len(buf))
        finally:
            assert L.cot_set_tuning(26, 0) == 0
        assert rc == 0, L.cot_last_error().decode()
        got = [_short(ln) for ln in buf.value.decode().splitlines() if ln]
        assert got == TABLE[self.key], (self.key, got, TABLE[self.key])
        rc = fn()
        assert rc == 0, L.cot_last_error().decode()
        torch.cuda.synchronize()


@pytest.fixture(autouse=True)
def _bench_tuning():
    """the state bench.py's `new` set runs in (KERNEL_SETS: BatchNorm finalize folded into the apply kernels)"""
    L = _lib.lib()
    assert L.cot_set_tuning(12, 1) == 0
    yield
    assert L.cot_set_tuning(12, 1) == 0


def _conv1x1g(key, shp, what):
    """grouped 1x1 (CoXtLayer, groups = 2): cot_conv1x1g_* -- group by group on the tuned kernels or the general kernel"""
    N, Ci, Co, g, H, W, s, bias = shp
    assert s == 1
    HW = H * W
    L, pin = _lib.lib(), _Pinned(key)
    seed = Ci * 5 + Co + HW
    x = _randn(N, Ci, H, W, seed=seed)
    w = _randn(Co, Ci // g, 1, 1, seed=seed + 1, scale=(Ci // g) ** -0.5)
    b = _randn(Co, seed=seed + 2) if bias else None
    gy = _randn(N, Co, H, W, seed=seed + 3)
    ws = torch.empty(max(int(L.cot_convg_workspace(N, Ci, Co, g, HW, 1, 1)), 256), dtype=torch.uint8, device=DEV)
    ws.fill_(0xFF)
    st = _st()
    if what == "fwd":
        y = torch.full((N, Co, H, W), float("nan"), device=DEV).bfloat16()
        pin.issue(lambda: L.cot_conv1x1g_forward(P(x), P(w), P(b), P(y), N, Ci, Co, g, HW, BF, st))
        _close(y, F.conv2d(x.float(), w.float(), b.float() if bias else None, groups=g))
    elif what == "dgrad":
        gx = torch.full((N, Ci, H, W), float("nan"), device=DEV).bfloat16()
        pin.issue(lambda: L.cot_conv1x1g_backward_data(P(gy), P(w), P(gx), 0, N, Ci, Co, g, HW, BF, st))
        _close(gx, F.conv_transpose2d(gy.float(), w.float(), None, groups=g))
    else:
        gw = torch.full((Co, Ci // g, 1, 1), float("nan"), device=DEV).bfloat16()
        gb = torch.full((Co,), float("nan"), device=DEV).bfloat16() if bias else None
        pin.issue(lambda: L.cot_conv1x1g_backward_weight(P(gy), P(x), P(gw), P(gb), P(ws), N, Ci, Co, g, HW, BF, st))
        _close(gw, torch.nn.grad.conv2d_weight(x.float(), (Co, Ci // g, 1, 1), gy.float(), groups=g))
        if bias:
            _close(gb, gy.float().sum((0, 2, 3)))


def _conv1x1(key, shp, what):
    N, Ci, Co, g, H, W, s, bias = shp
    if g != 1:
        return _conv1x1g(key, shp, what)
    if s != 1:
        H, W = (H - 1) // s + 1, (W - 1) // s + 1
    HW = H * W
    L, pin = _lib.lib(), _Pinned(key)
    seed = Ci * 7 + Co + HW
    x = _randn(N, Ci, H, W, seed=seed)
    w = _randn(Co, Ci, 1, 1, seed=seed + 1, scale=Ci ** -0.5)
    b = _randn(Co, seed=seed + 2) if bias else None
    gy = _randn(N, Co, H, W, seed=seed + 3)
    ws = torch.empty(max(int(L.cot_conv1x1_workspace(N, Ci, Co, HW, int(bias))), 256), dtype=torch.uint8, device=DEV)
    ws.fill_(0xFF)  # (poisoned: partial sums must be written before they are read)
    st = _st()
    if what in ("fwd", "fwd(two slabs)"):
        y = torch.full((N, Co, H, W), float("nan"), device=DEV).bfloat16()
        if what == "fwd":
            pin.issue(lambda: L.cot_conv1x1_forward(P(x), None, Ci, P(w), P(b), P(y), N, Ci, Co, HW, BF, st))
        else:
            x1, x2 = x[:, :Ci // 2].contiguous(), x[:, Ci // 2:].contiguous()
            pin.issue(lambda: L.cot_conv1x1_forward(P(x1), P(x2), Ci // 2, P(w), None, P(y), N, Ci, Co, HW, BF, st))
        _close(y, F.conv2d(x.float(), w.float(), b.float() if bias else None))
    elif what == "dgrad":
        gx = torch.full((N, Ci, H, W), float("nan"), device=DEV).bfloat16()
        pin.issue(lambda: L.cot_conv1x1_backward_data(P(gy), P(w), P(gx), None, Ci, 0, P(ws), N, Ci, Co, HW, BF, st))
        _close(gx, torch.einsum("oc,nohw->nchw", w.float()[:, :, 0, 0], gy.float()))
    else:
        gw = torch.full((Co, Ci, 1, 1), float("nan"), device=DEV).bfloat16()
        gb = torch.full((Co,), float("nan"), device=DEV).bfloat16() if bias else None
        pin.issue(lambda: L.cot_conv1x1_backward_weight(P(gy), P(x), None, Ci, P(gw), P(gb), P(ws), N, Ci, Co, HW, BF, st))
        _close(gw[:, :, 0, 0], torch.einsum("nohw,nchw->oc", gy.float(), x.float()))
        if bias:
            _close(gb, gy.float().sum((0, 2, 3)))


def _stem3x3(key, shp, what):
    """a deep stem's first convolution (3 x 3 / stride 2, 3 -> Co: csrc/stem3x3.hip) at the table's size"""
    N, Ci, Co, G, H, W, s, bias = shp
    L, pin = _lib.lib(), _Pinned(key)
    x = _randn(N, 3, H, W, seed=H + Co)
    w = _randn(Co, 3, 3, 3, seed=H + Co + 1, scale=27 ** -0.5)
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    st = _st()
    if what == "fwd":
        y = torch.full((N, Co, Ho, Wo), float("nan"), device=DEV).bfloat16()
        pin.issue(lambda: L.cot_stem3x3s2_forward(P(x), P(w), P(y), N, H, W, Co, BF, st))
        _close(y, F.conv2d(x.float(), w.float(), None, 2, 1))
    else:
        gy = _randn(N, Co, Ho, Wo, seed=H + Co + 2)
        ws = torch.empty(int(L.cot_stem3x3s2_workspace(N, H, W, Co)), dtype=torch.uint8, device=DEV)
        ws.fill_(0xFF)
        gw = torch.full((Co, 3, 3, 3), float("nan"), device=DEV).bfloat16()
        pin.issue(lambda: L.cot_stem3x3s2_backward_weight(P(gy), P(x), P(gw), P(ws), N, H, W, Co, BF, st))
        _close(gw, torch.nn.grad.conv2d_weight(x.float(), (Co, 3, 3, 3), gy.float(), 2, 1))


def _conv3x3(key, shp, what):
    N, Ci, Co, G, H, W, s, bias = shp
    if s == 2 and Ci == 3:
        return _stem3x3(key, shp, what)
    L, pin = _lib.lib(), _Pinned(key)
    seed = Ci * 3 + H
    HW = H * W
    lead = (W + 1 + 7) // 8 * 8
    flat = torch.full((N * Ci * HW + 2 * lead,), float("nan"), device=DEV).bfloat16()  # NaN margins (the guarded kernel reads them)
    x = flat[lead:lead + N * Ci * HW].view(N, Ci, H, W)
    x.copy_(_randn(N, Ci, H, W, seed=seed))
    w = _randn(Co, Ci // G, 3, 3, seed=seed + 1, scale=(9 * Ci // G) ** -0.5)
    gy = _randn(N, Co, H, W, seed=seed + 2)
    masks = torch.empty(int(L.cot_conv3x3g_masks_bytes(H, W)), dtype=torch.uint8, device=DEV)
    st = _st()
    assert L.cot_conv3x3g_masks(P(masks), H, W, st) == 0
    # groups of 12 channels: what the nodes launch is pairs of groups as ONE group of 24 with a block-diagonal weight (cot_layer_fused.
    # _merged_weight, made by the product's own helper here) -- compared with torch on the module's grouping
    from cotnet_amd import cot_layer_fused as clf
    Gl, wl = G, w
    if Ci == Co and clf._merge12(Ci, G) and what in ("fwd", "dgrad"):
        conv = torch.nn.Conv2d(Ci, Co, 3, 1, 1, groups=G, bias=False).to(DEV).bfloat16()
        with torch.no_grad():
            conv.weight.copy_(w)
        Gl, wl = G // 2, clf._merged_weight(conv, Ci, G, True)
    ws = torch.empty(max(int(L.cot_conv3x3g_workspace(N, Ci, Co, G, H, W)), int(L.cot_conv3x3g_workspace(N, Ci, Co, Gl, H, W)), 256),
                     dtype=torch.uint8, device=DEV)
    ws.fill_(0xFF)
    if what == "fwd":
        y = torch.full((N, Co, H, W), float("nan"), device=DEV).bfloat16()
        pin.issue(lambda: L.cot_conv3x3g_forward(P(x), P(wl), P(y), P(masks), P(ws), N, Ci, Co, Gl, H, W, BF, st))
        _close(y, F.conv2d(x.float(), w.float(), None, 1, 1, 1, G))
    elif what == "dgrad":
        gx = torch.full((N, Ci, H, W), float("nan"), device=DEV).bfloat16()
        pin.issue(lambda: L.cot_conv3x3g_backward_data(P(gy), P(wl), P(gx), 0, P(masks), P(ws), N, Ci, Co, Gl, H, W, BF, st))
        _close(gx, F.conv_transpose2d(gy.float(), w.float(), None, 1, 1, 0, G))
    else:
        gw = torch.full((Co, Ci // G, 3, 3), float("nan"), device=DEV).bfloat16()
        if what == "wgrad":
            pin.issue(lambda: L.cot_conv3x3g_backward_weight(P(gy), P(x), P(gw), P(masks), P(ws), N, Ci, Co, G, H, W, BF, st))
        else:
            pin.issue(lambda: L.cot_conv3x3g_backward_weight_guarded(P(gy), P(x), P(gw), P(masks), P(ws), N, Ci, Co, G, H, W, BF,
                                                                      (W + 8) // 8 * 8, st))
        ref = torch.nn.grad.conv2d_weight(x.float(), (Co, Ci // G, 3, 3), gy.float(), 1, 1, 1, G)
        _close(gw, ref)


def _bn(key, shp, what):
    N, C, HW = shp
    L, pin = _lib.lib(), _Pinned(key)
    seed = C + HW
    x = _randn(N, C, HW, 1, seed=seed) * 1.5 + 0.25
    x = x.bfloat16()
    gamma, beta = _gamma_beta(C, seed)
    mean, rstd = torch.empty(C, device=DEV), torch.empty(C, device=DEV)
    rm, rv = torch.zeros(C, device=DEV), torch.ones(C, device=DEV)
    nbt = torch.zeros((), dtype=torch.int64, device=DEV)
    nws = int(L.cot_bn_act_workspace(N, C))
    ws = torch.full((max(nws, 1),), float("nan"), device=DEV)
    y = torch.full((N, C, HW, 1), float("nan"), device=DEV).bfloat16()
    st = _st()
    fwd = lambda: L.cot_bn_act_forward(P(x), None, P(y), P(gamma), P(beta), P(mean), P(rstd), P(rm), P(rv), P(nbt), P(ws),  # noqa: E731
                                       N, C, HW, 1e-5, 0.1, 1, BF, st)
    xr = x.float().requires_grad_(True)
    gr, br = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    rm_r, rv_r = torch.zeros(C, device=DEV), torch.ones(C, device=DEV)
    yr = F.relu(F.batch_norm(xr, rm_r, rv_r, gr, br, True, 0.1, 1e-5))
    if what == "fwd":
        pin.issue(fwd)
        _close(y, yr.detach())
        _rel(mean, xr.detach().mean((0, 2, 3)), 1e-4)
        _rel(rm, rm_r, 1e-3)
        _rel(rv, rv_r, 1e-3)
        assert int(nbt.item()) == 1  # (the dry run touches nothing; one real call)
        return
    assert fwd() == 0
    dy = _randn(N, C, HW, 1, seed=seed + 1)
    dx = torch.full((N, C, HW, 1), float("nan"), device=DEV).bfloat16()
    dg, db = torch.full((C,), float("nan"), device=DEV), torch.full((C,), float("nan"), device=DEV)
    ws2 = torch.full((max(nws, 1),), float("nan"), device=DEV)
    pin.issue(lambda: L.cot_bn_act_backward(P(dy), P(x), None, P(dx), None, P(gamma), P(beta), P(mean), P(rstd), P(dg), P(db),
                                            P(ws2), N, C, HW, 1, BF, st))
    # the kernel's ReLU mask is the sign of ITS pre-activation; where |pre-activation| is below the bf16 resolution of the
    # inputs the fp32 reference may decide differently: compare where the reference's pre-activation is clearly off zero
    yr.backward(dy.float())
    pre = F.batch_norm(x.float(), None, None, gamma, beta, True, 0.0, 1e-5)
    clear = pre.abs() > 1e-2
    d = (dx.float() - xr.grad).abs()
    tol = 1e-2 * (xr.grad.abs() + xr.grad.abs().mean())
    assert (d[clear] <= tol[clear]).all(), d[clear].max().item()
    assert (~clear).float().mean().item() < 0.02
    dg_r, db_r = _param_grads_under_the_kernels_mask(x, dy.float(), y)
    _rel(dg, dg_r, 5e-3 if N * HW > 1000 else 2e-2)
    _rel(db, db_r, 5e-3 if N * HW > 1000 else 2e-2)


def _gn(key, shp, what):
    N, C, HW, G = shp
    L, pin = _lib.lib(), _Pinned(key)
    seed = C + HW
    x = _randn(N, C, HW, 1, seed=seed)
    gamma, beta = (t.bfloat16() for t in _gamma_beta(C, C + HW))
    mean, rstd = torch.empty(N * G, device=DEV), torch.empty(N * G, device=DEV)
    y = torch.full((N, C, HW, 1), float("nan"), device=DEV).bfloat16()
    st = _st()
    fwd = lambda: L.cot_group_norm9_forward(P(x), P(gamma), P(beta), P(y), P(mean), P(rstd), N, C, HW, 1e-5, BF, st)  # noqa: E731
    xr = x.float().requires_grad_(True)
    gr, br = gamma.float().requires_grad_(True), beta.float().requires_grad_(True)
    yr = F.group_norm(xr, G, gr, br, 1e-5)
    if what == "fwd":
        pin.issue(fwd)
        _close(y, yr.detach())
        _rel(mean.view(N, G), xr.detach().view(N, G, -1).mean(-1), 1e-3)
        return
    assert fwd() == 0
    dy = _randn(N, C, HW, 1, seed=seed + 1)
    dx = torch.full((N, C, HW, 1), float("nan"), device=DEV).bfloat16()
    dg, db = torch.full((C,), float("nan"), device=DEV).bfloat16(), torch.full((C,), float("nan"), device=DEV).bfloat16()
    ws = torch.full((2 * N * C,), float("nan"), device=DEV)
    pin.issue(lambda: L.cot_group_norm9_backward(P(dy), P(x), P(mean), P(rstd), P(gamma), P(dx), P(dg), P(db), P(ws), N, C, HW,
                                                 BF, st))
    yr.backward(dy.float())
    _close(dx, xr.grad)
    _close(dg, gr.grad)
    _close(db, br.grad)


def _agg(key, shp, what):
    from oracle import cref
    N, C, H, W, wC = shp
    L, pin = _lib.lib(), _Pinned(key)
    geom = _lib.AggGeom(N, C, H, W, 1, wC, 3, 3, 1, 1, 1, 1, 1, 1)
    st = _st()
    for integer in (False, True):
        g = torch.Generator().manual_seed(C + H + int(integer))
        if integer:  # small integers: every product and partial sum is exact in bf16 / fp32 -> bit-exact bookkeeping
            x = torch.randint(-3, 4, (N, C, H, W), generator=g).float()
            w = torch.randint(-2, 3, (N, 1, wC, 9, H, W), generator=g).float()
            go = torch.randint(-2, 3, (N, C, H, W), generator=g).float()
        else:
            x = torch.randn(N, C, H, W, generator=g).bfloat16().float()
            w = torch.randn(N, 1, wC, 9, H, W, generator=g).bfloat16().float()
            go = torch.randn(N, C, H, W, generator=g).bfloat16().float()
        xd, wd, gd = x.to(DEV).bfloat16(), w.to(DEV).bfloat16(), go.to(DEV).bfloat16()
        if what == "fwd":
            out = torch.full((N, C, H, W), float("nan"), device=DEV).bfloat16()
            pin.issue(lambda: L.cot_agg_forward(P(xd), P(wd), P(out), ctypes.byref(geom), BF, 0, st))
            ref = cref.forward(x, w, 3, 1, 1, 1)
            pairs = [(out, ref)]
        else:
            gx = torch.full((N, C, H, W), float("nan"), device=DEV).bfloat16()
            gw = torch.full((N, 1, wC, 9, H, W), float("nan"), device=DEV).bfloat16()
            pin.issue(lambda: L.cot_agg_backward(P(gd), P(xd), P(wd), P(gx), P(gw), ctypes.byref(geom), BF, 0, st))
            pairs = [(gx, cref.backward_input(go, w, x.shape, 3, 1, 1, 1)), (gw, cref.backward_weight(go, x, w.shape, 3, 1, 1, 1))]
        for got, ref in pairs:
            got = got.float().cpu()
            if integer:
                assert torch.equal(got, ref), (got - ref).abs().max().item()
            else:  # fp32 sums of <= 72 exact products, one bf16 rounding of the result
                assert ((got - ref).abs() <= 2.0 ** -8 * ref.abs() + 1e-5 * (1 + ref.abs().mean())).all(), (got - ref).abs().max().item()


def _pool(key, kind, shp, what):
    N, C, H, W = shp
    L, pin = _lib.lib(), _Pinned(key)
    x = _randn(N, C, H, W, seed=C + H)
    if kind == "MaxPool2d":
        x = F.relu(x)  # (after the stem's ReLU: ties at zero are the common case)
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    y = torch.full((N, C, Ho, Wo), float("nan"), device=DEV).bfloat16()
    st = _st()
    if kind == "AvgPool2d":
        pin.issue(lambda: L.cot_avgpool3x3s2_forward(P(x), P(y), N * C, H, W, BF, st))
        _close(y, F.avg_pool2d(x.float(), 3, 2, 1), 5e-3)
    else:
        pin.issue(lambda: L.cot_maxpool3x3s2_forward(P(x), P(y), N * C, H, W, BF, st))
        assert torch.equal(y, F.max_pool2d(x, 3, 2, 1))


def _bn_lay(key, shp, what):
    """the per-tensor-layout BatchNorm of the channel-major blocks (cot_bn_act_*_lay) in the table's representative form -- forward:
    x channel-major, y NCHW + y2 channel-major (bn1); backward: dy and x channel-major, dy2 NCHW, dx channel-major (its mirror) --
    against fp32 torch on the un-permuted tensors"""
    N, C, HW = shp
    L, pin = _lib.lib(), _Pinned(key)
    seed = C + HW
    x = (_randn(N, C, HW, 1, seed=seed).float() * 1.5 + 0.25).bfloat16()
    gamma, beta = _gamma_beta(C, seed)
    mean, rstd = torch.empty(C, device=DEV), torch.empty(C, device=DEV)
    rm, rv = torch.zeros(C, device=DEV), torch.ones(C, device=DEV)
    nbt = torch.zeros((), dtype=torch.int64, device=DEV)
    cmj = lambda t: t.transpose(0, 1).contiguous()  # noqa: E731
    xc = cmj(x)
    y = torch.full((N, C, HW, 1), float("nan"), device=DEV).bfloat16()
    y2 = torch.full((C, N, HW, 1), float("nan"), device=DEV).bfloat16()
    st = _st()
    fwd = lambda: L.cot_bn_act_forward_lay(P(xc), None, P(y), P(y2), P(gamma), P(beta), P(mean), P(rstd), P(rm), P(rv), P(nbt), None,  # noqa: E731
                                           N, C, HW, 1e-5, 0.1, 1, 1 | 8, BF, st)
    xr = x.float().requires_grad_(True)
    gr, br = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    rm_r, rv_r = torch.zeros(C, device=DEV), torch.ones(C, device=DEV)
    yr = F.relu(F.batch_norm(xr, rm_r, rv_r, gr, br, True, 0.1, 1e-5))
    if what == "fwd":
        pin.issue(fwd)
        _close(y, yr.detach())
        assert torch.equal(cmj(y2), y)
        _rel(mean, xr.detach().mean((0, 2, 3)), 1e-4)
        _rel(rm, rm_r, 1e-3)
        _rel(rv, rv_r, 1e-3)
        assert int(nbt.item()) == 1
        return
    assert fwd() == 0
    dy, dy2 = _randn(N, C, HW, 1, seed=seed + 1), _randn(N, C, HW, 1, seed=seed + 2)
    dyc = cmj(dy)
    dxc = torch.full((C, N, HW, 1), float("nan"), device=DEV).bfloat16()
    dg, db = torch.full((C,), float("nan"), device=DEV), torch.full((C,), float("nan"), device=DEV)
    pin.issue(lambda: L.cot_bn_act_backward_lay(P(dyc), P(dy2), P(xc), None, P(dxc), None, P(gamma), P(beta), P(mean), P(rstd), P(dg), P(db),
                                                None, N, C, HW, 1, 1 | 4 | 16, BF, st))
    dsum = (dy.float() + dy2.float()).bfloat16().float()  # (the kernel: fp32 sum of the two contributions, one rounding)
    yr.backward(dsum)
    pre = F.batch_norm(x.float(), None, None, gamma, beta, True, 0.0, 1e-5)
    clear = pre.abs() > 1e-2
    d = (cmj(dxc).float() - xr.grad).abs()
    tol = 1e-2 * (xr.grad.abs() + xr.grad.abs().mean())
    assert (d[clear] <= tol[clear]).all(), d[clear].max().item()
    assert (~clear).float().mean().item() < 0.02
    dg_r, db_r = _param_grads_under_the_kernels_mask(x, dsum, y)
    _rel(dg, dg_r, 5e-3)
    _rel(db, db_r, 5e-3)


def _run_entry(key):
    _, kind, shape, what = key.split(" ", 3)
    shp = tuple(int(v) for v in shape.split("x"))
    if kind == "conv1x1cm":    # channel rows: ONE image of N*H*W pixels (cot_layer_fused._BottleneckCMNode)
        N, Ci, Co, g, H, W, s, bias = shp
        return _conv1x1(key, (1, Ci, Co, 1, 1, N * H * W, 1, bias), what)
    if kind == "bncm":
        return _bn(key, (1, shp[1], shp[0] * shp[2]), what)
    if kind == "bnlay":
        return _bn_lay(key, shp, what)
    if kind == "conv1x1":
        _conv1x1(key, shp, what)
    elif kind == "conv3x3":
        _conv3x3(key, shp, what)
    elif kind == "bn":
        _bn(key, shp, what)
    elif kind == "gn":
        _gn(key, shp, what)
    elif kind == "agg":
        _agg(key, shp, what)
    elif kind in ("AvgPool2d", "MaxPool2d"):
        _pool(key, kind, shp, what)
    else:
        raise AssertionError(f"no parity case for table entry {key!r}")


@pytest.mark.parametrize("key", KEYS, ids=[k[len(CFG) + 1:].replace(" ", "-") for k in KEYS])
def test_headline_dispatch_entry(key):
    _run_entry(key)


@pytest.mark.parametrize("key", OTHER_KEYS, ids=[k.replace(" ", "-") for k in OTHER_KEYS])
def test_secondary_config_dispatch_entry(key):
    """CoTNeXt-101 2x48d (B = 64) and SE-CoTNetD-152 at 320 x 320 (B = 64): grouped 1x1 convolutions group by group on the tuned
    kernels, groups-8 / dense 3x3 convolutions on the LDS kernels (K padding, row blocks, 160-pixel rows), the N' = 2B fold of the
    aggregation, the 40 / 20 / 10-pixel planes"""
    _run_entry(key)


def test_every_headline_entry_has_a_case():
    kinds = {k.split(" ")[1] for k in KEYS}
    assert kinds <= {"conv1x1", "conv3x3", "bn", "gn", "agg", "AvgPool2d", "MaxPool2d", "conv1x1cm", "bncm", "bnlay"}, kinds
    assert len(KEYS) >= 240
    assert {k.split(" ")[1] for k in OTHER_KEYS} <= {"conv1x1", "conv3x3", "bn", "gn", "agg", "AvgPool2d", "MaxPool2d"}
    assert len(OTHER_KEYS) >= 380

"""GPU parity tests of the HIP aggregation kernels (through the C ABI, via cotnet_amd.aggregation_zeropad).

Checker = oracle/ (C restatement pinned bit-for-bit to the reference's kernels) + the committed reference fixtures.
Bars: fp64 1e-9 (the reference's own self-test threshold), fp32 1e-5 (BASELINE bar is 1e-3), integer-valued data
bit-exact, padded-tap weight gradients exactly 0, bf16/fp16 against the fp32 oracle on rounded inputs.
"""
import numpy as np
import pytest
import torch

from cotnet_amd import _lib
from cotnet_amd.aggregation_zeropad import LocalConvolution, aggregation_zeropad
from cotnet_amd.aggregation_zeropad_mix import AggregationZeropadMix, LocalConvolutionMix, aggregation_zeropad_mix
from oracle import cref, unfold_oracle
from tests.conftest import AGG_FIXTURES, MIX_FIXTURES, agg_case_inputs, load_golden, mix_case_inputs

pytestmark = pytest.mark.gpu
DEV = "cuda"


def run_hip(x, w, gout, k, s, p, d, layout="nchw"):
    """forward + both grads on the GPU; tensors given as CPU NCHW"""
    xg = x.to(DEV).requires_grad_(True)
    wg = w.to(DEV).requires_grad_(True)
    if layout == "nhwc":
        xg = xg.detach().contiguous(memory_format=torch.channels_last).requires_grad_(True)
        N, heads, wC, taps, Ho, Wo = w.shape
        wl = w.to(DEV).permute(0, 4, 5, 1, 2, 3).contiguous().permute(0, 3, 4, 5, 1, 2)
        assert wl.shape == w.shape
        wg = wl.detach().requires_grad_(True)
    y = aggregation_zeropad(xg, wg, k, s, p, d)
    fwd_kernel = _lib.last_kernel()
    y.backward(gout.to(DEV))
    bwd_kernel = _lib.last_kernel()
    torch.cuda.synchronize()
    return y.detach().cpu(), xg.grad.cpu(), wg.grad.cpu(), fwd_kernel, bwd_kernel


def oracle_all(x, w, gout, k, s, p, d):
    return (cref.forward(x, w, k, s, p, d), cref.backward_input(gout, w, x.shape, k, s, p, d),
            cref.backward_weight(gout, x, w.shape, k, s, p, d))


@pytest.mark.parametrize("layout", ["nchw", "nhwc"])
@pytest.mark.parametrize("name", AGG_FIXTURES)
def test_reference_fixtures(name, layout):
    gold = load_golden(name)
    geom, x, w, gout = agg_case_inputs(gold)
    k, s, p, d = geom["kernel_size"], geom["stride"], geom["padding"], geom["dilation"]
    y, gx, gw, _, _ = run_hip(x, w, gout, k, s, p, d, layout)
    tol = 1e-9 if x.dtype == torch.float64 else 1e-5
    assert (y - torch.from_numpy(gold["out"])).abs().max() < tol
    assert (gx - torch.from_numpy(gold["gx"])).abs().max() < tol
    assert (gw - torch.from_numpy(gold["gw"])).abs().max() < tol


STAGE_SHAPES = [  # CoTNet-50 CoT-layer geometries (C, H=W), small batch
    (64, 56), (128, 28), (256, 14), (512, 7),
]


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
@pytest.mark.parametrize("layout", ["nchw", "nhwc"])
@pytest.mark.parametrize("C,HW", STAGE_SHAPES)
def test_k3_fast_path_stage_shapes(C, HW, layout, dtype):
    g = torch.Generator().manual_seed(C + HW)
    N, wC = 3, C // 8
    x = torch.randn(N, C, HW, HW, dtype=dtype, generator=g)
    w = torch.randn(N, 1, wC, 9, HW, HW, dtype=dtype, generator=g)
    gout = torch.randn(N, C, HW, HW, dtype=dtype, generator=g)
    y, gx, gw, fk, bk = run_hip(x, w, gout, 3, 1, 1, 1, layout)
    oy, ogx, ogw = oracle_all(x, w, gout, 3, 1, 1, 1)
    tol = 1e-9 if dtype == torch.float64 else 2e-5
    assert (y - oy).abs().max() < tol and (gx - ogx).abs().max() < tol and (gw - ogw).abs().max() < tol
    if layout == "nchw":
        assert "k3" in fk and "k3" in bk, (fk, bk)  # the fast path is what ran


@pytest.mark.parametrize("layout", ["nchw", "nhwc"])
@pytest.mark.parametrize("k,s,p,d,heads,N,C,wC,H,W", [
    (3, 1, 1, 1, 2, 2, 16, 4, 9, 12),     # heads > 1 on the 3x3 geometry
    (5, 1, 2, 1, 2, 2, 8, 4, 9, 9),
    (1, 1, 0, 1, 2, 2, 8, 4, 9, 9),
    (3, 2, 1, 1, 1, 2, 8, 2, 11, 10),
    (3, 1, 2, 2, 1, 1, 8, 4, 11, 10),
    ((3, 5), (2, 1), (1, 2), (1, 1), 1, 1, 6, 3, 9, 12),
    (3, 1, 0, 1, 1, 2, 8, 8, 7, 9),       # no padding: output smaller than input
    (3, 3, 1, 2, 1, 1, 4, 1, 13, 13),     # stride 3 + dilation 2: exercises the truncating %,/ on negatives
    (3, 1, 1, 1, 1, 2, 24, 3, 5, 5),      # wC not a multiple of the vector width
    (3, 1, 1, 1, 1, 1, 8, 8, 1, 1),       # 1x1 image: every tap but the centre is padding
])
def test_generic_geometries_fp64(k, s, p, d, heads, N, C, wC, H, W, layout):
    g = torch.Generator().manual_seed(5)
    Ho, Wo = unfold_oracle.out_hw(H, W, k, s, p, d)
    kk = (k, k) if isinstance(k, int) else k
    x = torch.randn(N, C, H, W, dtype=torch.float64, generator=g)
    w = torch.randn(N, heads, wC, kk[0] * kk[1], Ho, Wo, dtype=torch.float64, generator=g)
    gout = torch.randn(N, heads * C, Ho, Wo, dtype=torch.float64, generator=g)
    y, gx, gw, _, _ = run_hip(x, w, gout, k, s, p, d, layout)
    oy, ogx, ogw = oracle_all(x, w, gout, k, s, p, d)
    assert (y - oy).abs().max() < 1e-9 and (gx - ogx).abs().max() < 1e-9 and (gw - ogw).abs().max() < 1e-9


@pytest.mark.parametrize("layout", ["nchw", "nhwc"])
@pytest.mark.parametrize("W", [56, 28, 14, 7, 10, 3])
def test_integer_data_bit_exact_and_padded_taps_zero(W, layout):
    g = torch.Generator().manual_seed(W)
    N, C, wC, H = 2, 16, 2, 6
    x = torch.randint(-4, 5, (N, C, H, W), generator=g).float()
    w = torch.randint(-3, 4, (N, 1, wC, 9, H, W), generator=g).float()
    gout = torch.randint(-2, 3, (N, C, H, W), generator=g).float()
    y, gx, gw, _, _ = run_hip(x, w, gout, 3, 1, 1, 1, layout)
    oy, ogx, ogw = oracle_all(x, w, gout, 3, 1, 1, 1)
    assert torch.equal(y, oy) and torch.equal(gx, ogx) and torch.equal(gw, ogw)
    assert torch.all(gw[:, :, :, 0, 0, :] == 0) and torch.all(gw[:, :, :, 0, :, 0] == 0)
    assert torch.all(gw[:, :, :, 8, -1, :] == 0) and torch.all(gw[:, :, :, 8, :, -1] == 0)


@pytest.mark.parametrize("dtype,tol", [(torch.bfloat16, 6e-2), (torch.float16, 8e-3)])
@pytest.mark.parametrize("layout", ["nchw", "nhwc"])
@pytest.mark.parametrize("C,HW", STAGE_SHAPES)
def test_half_precision_storage(C, HW, layout, dtype, tol):
    """no reference exists for bf16/fp16 (utils.py:8-12 rejects them): compare with the fp32 oracle evaluated on
    the rounded inputs; error budget = one output rounding (2^-8 resp. 2^-11 relative) of |sum| <~ 15."""
    g = torch.Generator().manual_seed(1)
    N, wC = 2, C // 8
    x = torch.randn(N, C, HW, HW, generator=g).to(dtype)
    w = torch.randn(N, 1, wC, 9, HW, HW, generator=g).to(dtype)
    gout = torch.randn(N, C, HW, HW, generator=g).to(dtype)
    y, gx, gw, fk, bk = run_hip(x, w, gout, 3, 1, 1, 1, layout)
    oy, ogx, ogw = oracle_all(x.float(), w.float(), gout.float(), 3, 1, 1, 1)
    for got, want in ((y, oy), (gx, ogx), (gw, ogw)):
        assert got.dtype == dtype
        err = (got.float() - want).abs()
        assert (err <= tol * (1.0 + want.abs())).all(), err.max()


def test_gradcheck_fp64():
    g = torch.Generator().manual_seed(9)
    x = torch.randn(2, 8, 5, 6, dtype=torch.float64, generator=g).to(DEV).requires_grad_(True)
    for k, p, heads in ((3, 1, 1), (5, 2, 2)):
        w = torch.randn(2, heads, 4, k * k, 5, 6, dtype=torch.float64, generator=g).to(DEV).requires_grad_(True)
        assert torch.autograd.gradcheck(lambda a, b: aggregation_zeropad(a, b, k, 1, p, 1), (x, w))


def test_module_surface_noncontiguous_and_cpu_route():
    g = torch.Generator().manual_seed(2)
    x = torch.randn(2, 16, 9, 8, generator=g)
    w = torch.randn(2, 1, 2, 9, 9, 8, generator=g)
    want = cref.forward(x, w, 3, 1, 1, 1)
    conv = LocalConvolution(16, 16, kernel_size=3, stride=1, padding=1, dilation=1)
    # CPU tensors: reference route (copy to GPU, compute, copy back)
    y = conv(x, w)
    assert y.device.type == "cpu" and (y - want).abs().max() < 1e-5
    # non-contiguous views (sliced channel dim / transposed storage)
    xb = torch.randn(2, 32, 9, 8, generator=g)
    xs = xb.to(DEV)[:, ::2]
    assert not xs.is_contiguous()
    ys = conv(xs, w.to(DEV))
    assert (ys.cpu() - cref.forward(xb[:, ::2].contiguous(), w, 3, 1, 1, 1)).abs().max() < 1e-5
    # needs_input_grad honoured: only w requires grad
    wg = w.to(DEV).requires_grad_(True)
    conv(x.to(DEV), wg).sum().backward()
    torch.cuda.synchronize()
    assert wg.grad is not None
    assert "gw" in _lib.last_kernel() and "gx" not in _lib.last_kernel(), _lib.last_kernel()
    with pytest.raises(AssertionError):
        conv(x.to(DEV), torch.randn(2, 1, 3, 9, 9, 8, device=DEV))  # 16 % 3 != 0


# ---- size-independent properties at the BASELINE per-GPU batch (B=80) ------------------------------------
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("layout", ["nchw", "nhwc"])
@pytest.mark.parametrize("C,HW", STAGE_SHAPES)
def test_full_size_properties(C, HW, layout, dtype):
    N, wC = 80, C // 8
    torch.manual_seed(0)
    mf = torch.channels_last if layout == "nhwc" else torch.contiguous_format

    def mk_w(t):
        if layout == "nhwc":
            return t.permute(0, 4, 5, 1, 2, 3).contiguous().permute(0, 3, 4, 5, 1, 2)
        return t

    x1 = torch.randn(N, C, HW, HW, device=DEV, dtype=dtype).contiguous(memory_format=mf)
    w = mk_w(torch.randn(N, 1, wC, 9, HW, HW, device=DEV, dtype=dtype))
    # (1) identity weights: centre tap 1, others 0 -> output == input, bit for bit
    wid = torch.zeros(N, 1, wC, 9, HW, HW, device=DEV, dtype=dtype)
    wid[:, :, :, 4] = 1
    assert torch.equal(aggregation_zeropad(x1, mk_w(wid), 3, 1, 1, 1), x1)
    # (2) shift weights: tap (0,0)=1 -> output[h,w] = x[h-1,w-1] with zero fill (pad bookkeeping at full size)
    wsh = torch.zeros(N, 1, wC, 9, HW, HW, device=DEV, dtype=dtype)
    wsh[:, :, :, 0] = 1
    ysh = aggregation_zeropad(x1, mk_w(wsh), 3, 1, 1, 1)
    want = torch.zeros_like(x1)
    want[:, :, 1:, 1:] = x1[:, :, :-1, :-1]
    assert torch.equal(ysh, want)
    # (3) adjointness of the three kernels:  <agg(x,w), g> == <x, gx> == <w, gw>   (fp64 accumulation of the dots)
    xr, wr = x1.clone().requires_grad_(True), w.detach().clone().requires_grad_(True)
    y = aggregation_zeropad(xr, wr, 3, 1, 1, 1)
    gout = torch.randn_like(y)
    y.backward(gout)
    lhs = (y.double() * gout.double()).sum()
    rx = (xr.grad.double() * x1.double()).sum()
    rw = (wr.grad.double() * w.double()).sum()
    scale = (y.double().abs() * gout.double().abs()).sum()
    rel = 1e-5 if dtype == torch.float32 else 4e-3
    assert abs(lhs - rx) < rel * scale and abs(lhs - rw) < rel * scale
    # (4) linearity in x (exact in fp32 for a 2x scaling)
    assert torch.equal(aggregation_zeropad(x1 * 2, w, 3, 1, 1, 1), y.detach() * 2)


# ---- mix variant -----------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", MIX_FIXTURES)
def test_mix_reference_fixtures(name):
    gold = load_golden(name)
    geom, x, w1, w2, gout = mix_case_inputs(gold)
    mod = LocalConvolutionMix(geom["C"], geom["C"], 3, 5, stride=1, padding1=1, padding2=2, dilation=1)
    xg, w1g, w2g = (t.to(DEV).requires_grad_(True) for t in (x, w1, w2))
    y = mod(xg, w1g, w2g)
    y.backward(gout.to(DEV))
    assert (y.detach().cpu() - torch.from_numpy(gold["out"])).abs().max() < 1e-9
    assert (xg.grad.cpu() - torch.from_numpy(gold["gx"])).abs().max() < 1e-9   # head-0-only quirk reproduced
    assert (w1g.grad.cpu() - torch.from_numpy(gold["gw1"])).abs().max() < 1e-9
    assert (w2g.grad.cpu() - torch.from_numpy(gold["gw2"])).abs().max() < 1e-9


def test_mix_all_heads_mode_is_the_true_gradient():
    gold = load_golden("agg_mix_heads2")
    geom, x, w1, w2, gout = mix_case_inputs(gold)
    AggregationZeropadMix.all_heads = True
    try:
        xg = x.to(DEV).requires_grad_(True)
        aggregation_zeropad_mix(xg, w1.to(DEV), w2.to(DEV), 3, 5, 1, 1, 2, 1).backward(gout.to(DEV))
    finally:
        AggregationZeropadMix.all_heads = False
    xr = x.clone().requires_grad_(True)
    unfold_oracle.aggregation_mix_unfold(xr, w1, w2, 1, 1, 2, 1).backward(gout)
    assert (xg.grad.cpu() - xr.grad).abs().max() < 1e-9


def _mix_gpu(x, w1, w2, gout, all_heads=False):
    """forward + the three gradients on the GPU through the autograd Function (= the C ABI); CPU tensors in and out"""
    xg, w1g, w2g = (t.to(DEV).requires_grad_(True) for t in (x, w1, w2))
    AggregationZeropadMix.all_heads = all_heads
    try:
        y = aggregation_zeropad_mix(xg, w1g, w2g, 3, 5, 1, 1, 2, 1)
        fk = _lib.last_kernel()
        y.backward(gout.to(DEV))
        bk = _lib.last_kernel()
    finally:
        AggregationZeropadMix.all_heads = False
    torch.cuda.synchronize()
    return y.detach().cpu(), xg.grad.cpu(), w1g.grad.cpu(), w2g.grad.cpu(), fk, bk


def _mix_oracle(x, w1, w2, gout, all_heads=False):
    return (cref.mix_forward(x, w1, w2, 1, 1, 2, 1), cref.mix_backward_input(gout, w1, w2, x.shape, 1, 1, 2, 1, all_heads),
            *cref.mix_backward_weight(gout, x, w1.shape, w2.shape, 1, 1, 2, 1))


def _mix_config5_inputs(dtype, heads=1, integer=False, N=64):
    """the op-level shape BASELINE config 5 / SURVEY 8(d) name: (B = 64, C = 256, 20 x 20, wC = 32, heads = 1)"""
    g = torch.Generator().manual_seed(4 + heads)
    C, wC, H, W = 256, 32, 20, 20
    if integer:
        mk = lambda lo, hi, *shape: torch.randint(lo, hi, shape, generator=g).to(dtype)
        return (mk(-4, 5, N, C, H, W), mk(-3, 4, N, heads, wC, 9, H, W), mk(-3, 4, N, heads, wC, 25, H, W),
                mk(-2, 3, N, 2 * heads * C, H, W))
    mk = lambda *shape: torch.randn(*shape, generator=g).to(dtype)
    return mk(N, C, H, W), mk(N, heads, wC, 9, H, W), mk(N, heads, wC, 25, H, W), mk(N, 2 * heads * C, H, W)


def test_mix_config5_shape_fp32_forward_and_both_gradients():
    """full batch, every element of the four results against the C oracle (reference loop order): 1e-5"""
    x, w1, w2, gout = _mix_config5_inputs(torch.float32)
    *got, fk, bk = _mix_gpu(x, w1, w2, gout)
    assert fk == "aggmix_fwd_tile" and bk in ("aggmix_bwd_input_tile", "aggmix_bwd_weight_tile"), (fk, bk)
    for a, b in zip(got, _mix_oracle(x, w1, w2, gout)):
        assert (a - b).abs().max() < 1e-5 * (1 + b.abs().max())
    assert torch.all(got[2][:, :, :, 0, 0, :] == 0) and torch.all(got[3][:, :, :, 24, :, -1] == 0)  # padded taps: exact zeros


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_mix_config5_shape_integer_data_is_bit_exact(dtype):
    x, w1, w2, gout = _mix_config5_inputs(dtype, integer=True)
    *got, fk, bk = _mix_gpu(x, w1, w2, gout)
    f = torch.float32
    for a, b in zip(got, _mix_oracle(x.to(f), w1.to(f), w2.to(f), gout.to(f))):
        assert torch.equal(a.float(), b)


def test_mix_config5_shape_bf16_storage_is_one_rounding_from_the_fp32_oracle():
    x, w1, w2, gout = _mix_config5_inputs(torch.bfloat16)
    *got, fk, bk = _mix_gpu(x, w1, w2, gout)
    assert fk == "aggmix_fwd_tile", fk
    f = torch.float32
    for a, b in zip(got, _mix_oracle(x.to(f), w1.to(f), w2.to(f), gout.to(f))):
        assert ((a.float() - b).abs() <= 2.0 ** -8 * b.abs() + 1e-5).all()  # half an ulp of bf16 (8 significand bits) + fp32 sum noise


@pytest.mark.parametrize("all_heads", [False, True])
def test_mix_config5_shape_two_heads(all_heads):
    """heads = 2: the reference's input gradient sums head 0 only (mix.py:87-88, reproduced by default); `all_heads` = the complete one"""
    x, w1, w2, gout = _mix_config5_inputs(torch.float32, heads=2, N=16)
    *got, fk, bk = _mix_gpu(x, w1, w2, gout, all_heads)
    want = _mix_oracle(x, w1, w2, gout, all_heads)
    for a, b in zip(got, want):
        assert (a - b).abs().max() < 1e-5 * (1 + b.abs().max())
    if all_heads:  # ... which is what autograd gives for the Unfold formula (a quarter of the batch: the formula is slow)
        xr = x[:4].clone().requires_grad_(True)
        unfold_oracle.aggregation_mix_unfold(xr, w1[:4], w2[:4], 1, 1, 2, 1).backward(gout[:4])
        assert (got[1][:4] - xr.grad).abs().max() < 1e-4
    else:
        assert (got[1] - _mix_oracle(x, w1, w2, gout, True)[1]).abs().max() > 0.1  # (the quirk is visible at two heads)


def test_mix_tile_kernels_equal_the_generic_kernels():
    """same sums in the same order: the LDS-tiled kernels and the one-lane-per-element kernels agree to fp32 rounding (the compiler
    contracts multiply-adds differently in the two loop shapes; integer-valued data is bit-exact for both, test above)"""
    x, w1, w2, gout = _mix_config5_inputs(torch.float32, N=8)
    tile = _mix_gpu(x, w1, w2, gout)
    _lib.lib().cot_set_tuning(51, 1)
    try:
        gen = _mix_gpu(x, w1, w2, gout)
    finally:
        _lib.lib().cot_set_tuning(51, 0)
    assert tile[4] == "aggmix_fwd_tile" and gen[4] == "aggmix_fwd"
    for a, b in zip(tile[:4], gen[:4]):
        assert (a - b).abs().max() <= 2e-6 * (1 + b.abs().max())


# ---- window softmax fused into the aggregation ------------------------------------------
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("C,HW", STAGE_SHAPES)
def test_fused_window_softmax(C, HW, dtype):
    from cotnet_amd.aggregation_zeropad import aggregation_zeropad_softmax
    g = torch.Generator().manual_seed(C)
    N, wC = 3, C // 8
    x = torch.randn(N, C, HW, HW, generator=g).to(dtype)
    logits = (torch.randn(N, 1, wC, 9, HW, HW, generator=g) * 2).to(dtype)
    gout = torch.randn(N, C, HW, HW, generator=g).to(dtype)
    xa, la = x.to(DEV).requires_grad_(True), logits.to(DEV).requires_grad_(True)
    y = aggregation_zeropad_softmax(xa, la, 3, 1, 1, 1)
    assert "softmax" in _lib.last_kernel()
    y.backward(gout.to(DEV))
    # checker: torch softmax + the ORACLE aggregation, differentiated by autograd, in fp32 on the same inputs
    xr, lr = x.float().requires_grad_(True), logits.float().requires_grad_(True)
    yr = unfold_oracle.aggregation_unfold(xr, torch.softmax(lr, dim=3), 3, 1, 1, 1)
    yr.backward(gout.float())
    tol = 2e-5 if dtype == torch.float32 else 4e-2
    assert ((y.detach().float().cpu() - yr.detach()).abs() <= tol * (1 + yr.detach().abs())).all()
    assert ((xa.grad.float().cpu() - xr.grad).abs() <= tol * (1 + xr.grad.abs())).all()
    assert ((la.grad.float().cpu() - lr.grad).abs() <= 4 * tol * (1 + lr.grad.abs())).all()


def test_window_softmax_composes_when_not_fusable():
    from cotnet_amd.aggregation_zeropad import aggregation_zeropad_softmax
    g = torch.Generator().manual_seed(3)
    x = torch.randn(2, 8, 9, 9, generator=g, dtype=torch.float64)
    logits = torch.randn(2, 2, 4, 25, 9, 9, generator=g, dtype=torch.float64)
    y = aggregation_zeropad_softmax(x.to(DEV), logits.to(DEV), 5, 1, 2, 1)   # 5x5, heads=2, fp64: composed path
    want = cref.forward(x, torch.softmax(logits, dim=3), 5, 1, 2, 1)
    assert (y.cpu() - want).abs().max() < 1e-9


def test_tensors_beyond_2_31_elements_index_correctly():
    """maximum sizes: N*C*H*W = 2.16e9 > 2^31 elements in every tensor (the reference's kernels index with `int`,
    cupy_layers/aggregation_zeropad.py:24-29, and would wrap here; a 288 GB device holds such a batch many times over).  Offsets
    past 2^31 are where the LAST images live: the first and the last images of the big batch must equal the same images run as
    a batch of their own, forward and both gradients (a wrapped offset reads another image: O(1) error)"""
    N, C, H, W = 10752, 64, 56, 56
    assert N * C * H * W > 2 ** 31
    torch.cuda.empty_cache()
    free, _ = torch.cuda.mem_get_info()
    if free < 48e9:
        pytest.skip("needs ~30 GB of device memory")
    g = torch.Generator(device=DEV).manual_seed(3)
    x = torch.randn(N, C, H, W, device=DEV, dtype=torch.bfloat16, generator=g).requires_grad_(True)
    w = torch.randn(N, 1, C // 8, 9, H, W, device=DEV, dtype=torch.bfloat16, generator=g).requires_grad_(True)
    gy = torch.randn(N, C, H, W, device=DEV, dtype=torch.bfloat16, generator=g)
    y = aggregation_zeropad(x, w, 3, 1, 1, 1)
    fk = _lib.last_kernel()
    y.backward(gy)
    bk = _lib.last_kernel()
    torch.cuda.synchronize()
    assert fk.startswith("agg_fwd_nchw_k3") and bk.startswith("agg_bwd_nchw_k3"), (fk, bk)
    for sl in (slice(0, 3), slice(N - 3, N)):
        xs, ws = x.detach()[sl].clone().requires_grad_(True), w.detach()[sl].clone().requires_grad_(True)
        ys = aggregation_zeropad(xs, ws, 3, 1, 1, 1)
        ys.backward(gy[sl].clone())
        torch.cuda.synchronize()
        for got, want in ((y.detach()[sl], ys.detach()), (x.grad[sl], xs.grad), (w.grad[sl], ws.grad)):
            want = want.float()
            assert (got.float() - want).abs().max().item() <= 2e-2 * want.abs().max().item(), (sl, fk, bk)

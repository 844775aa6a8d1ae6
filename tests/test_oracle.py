"""CPU tests that PIN the oracle (oracle/agg_oracle.c via oracle/cref.py).

Three anchors, strongest first:
  1. the committed fixtures in tests/golden/agg_*.npz -- outputs of the reference's own kernel source run on the
     CPU (tests/golden/make_golden.py): the C restatement must match them BIT FOR BIT (same summation order);
  2. when oracle/_ref holds prebuilt reference kernels (always in the build container, and on the GPU box because
     the .so files travel), the same comparison live on fresh random inputs;
  3. the analytic nn.Unfold formula of the reference's self-tests (aggregation_zeropad.py:249-251) at the
     reference's own 1e-9 fp64 threshold, forward and both gradients, on the self-test shapes.
"""
import json

import numpy as np
import pytest
import torch

from oracle import build_ref, cref, unfold_oracle
from tests.conftest import AGG_FIXTURES, MIX_FIXTURES, agg_case_inputs, load_golden, mix_case_inputs


@pytest.mark.parametrize("name", AGG_FIXTURES)
def test_c_oracle_matches_reference_fixture_bit_exact(name):
    gold = load_golden(name)
    geom, x, w, gout = agg_case_inputs(gold)
    k, s, p, d = geom["kernel_size"], geom["stride"], geom["padding"], geom["dilation"]
    assert torch.equal(cref.forward(x, w, k, s, p, d), torch.from_numpy(gold["out"]))
    assert torch.equal(cref.backward_input(gout, w, x.shape, k, s, p, d), torch.from_numpy(gold["gx"]))
    assert torch.equal(cref.backward_weight(gout, x, w.shape, k, s, p, d), torch.from_numpy(gold["gw"]))


@pytest.mark.parametrize("name", MIX_FIXTURES)
def test_c_oracle_mix_matches_reference_fixture_bit_exact(name):
    gold = load_golden(name)
    geom, x, w1, w2, gout = mix_case_inputs(gold)
    kw = dict(stride=geom["stride"], padding1=geom["padding1"], padding2=geom["padding2"], dilation=geom["dilation"])
    assert torch.equal(cref.mix_forward(x, w1, w2, **kw), torch.from_numpy(gold["out"]))
    assert torch.equal(cref.mix_backward_input(gout, w1, w2, x.shape, **kw), torch.from_numpy(gold["gx"]))
    gw1, gw2 = cref.mix_backward_weight(gout, x, w1.shape, w2.shape, **kw)
    assert torch.equal(gw1, torch.from_numpy(gold["gw1"])) and torch.equal(gw2, torch.from_numpy(gold["gw2"]))


def test_mix_reference_input_grad_is_head0_only():
    """documents the reference quirk (mix.py:87-88): with heads=2 the fixture's gx differs from the full gradient"""
    gold = load_golden("agg_mix_heads2")
    geom, x, w1, w2, gout = mix_case_inputs(gold)
    kw = dict(stride=1, padding1=1, padding2=2, dilation=1)
    full = cref.mix_backward_input(gout, w1, w2, x.shape, all_heads=True, **kw)
    xr = x.clone().requires_grad_(True)
    unfold_oracle.aggregation_mix_unfold(xr, w1, w2, **kw).backward(gout)
    assert (full - xr.grad).abs().max() < 1e-9
    assert (torch.from_numpy(gold["gx"]) - xr.grad).abs().max() > 1e-3


@pytest.mark.parametrize("idx", range(len(build_ref.PREBUILT_AGG)))
def test_c_oracle_matches_live_reference_kernels(idx):
    geom = build_ref.PREBUILT_AGG[idx]
    try:
        ref = build_ref.RefAggregation(**geom)
    except FileNotFoundError:
        pytest.skip("oracle/_ref not built and no reference checkout")
    dtype = torch.float32 if geom["dtype"] == "float" else torch.float64
    g = torch.Generator().manual_seed(100 + idx)
    x = torch.randn(geom["N"], geom["C"], geom["H"], geom["W"], dtype=dtype, generator=g)
    k = build_ref._pair(geom["kernel_size"])
    w = torch.randn(geom["N"], geom["heads"], geom["wC"], k[0] * k[1], ref.Ho, ref.Wo, dtype=dtype, generator=g)
    gout = torch.randn(geom["N"], geom["heads"] * geom["C"], ref.Ho, ref.Wo, dtype=dtype, generator=g)
    a = (geom["kernel_size"], geom["stride"], geom["padding"], geom["dilation"])
    assert torch.equal(ref.forward(x, w), cref.forward(x, w, *a))
    assert torch.equal(ref.backward_input(gout, w), cref.backward_input(gout, w, x.shape, *a))
    assert torch.equal(ref.backward_weight(gout, x), cref.backward_weight(gout, x, w.shape, *a))


SELFTEST_SHAPES = [
    # (k, s, d, p, heads, n, c_x, c_w, H, W): aggregation_zeropad.py:238-246 and :266-274
    (5, 1, 1, 2, 2, 2, 8, 4, 9, 9),
    (1, 1, 1, 0, 2, 2, 8, 4, 9, 9),
    (3, 1, 1, 1, 1, 2, 64, 8, 32, 32),   # BASELINE config 1
    (3, 2, 1, 1, 2, 1, 8, 2, 11, 10),
    (3, 1, 2, 2, 1, 1, 8, 4, 11, 10),
    (3, 1, 1, 0, 1, 2, 8, 8, 7, 9),      # padding 0: output smaller than input
]


@pytest.mark.parametrize("k,s,d,p,heads,n,c_x,c_w,H,W", SELFTEST_SHAPES)
def test_c_oracle_vs_unfold_selftest(k, s, d, p, heads, n, c_x, c_w, H, W):
    g = torch.Generator().manual_seed(7)
    Ho, Wo = unfold_oracle.out_hw(H, W, k, s, p, d)
    x = torch.randn(n, c_x, H, W, dtype=torch.float64, generator=g, requires_grad=True)
    w = torch.randn(n, heads, c_w, k * k, Ho, Wo, dtype=torch.float64, generator=g, requires_grad=True)
    y2 = unfold_oracle.aggregation_unfold(x, w, k, s, p, d)
    y1 = cref.forward(x.detach(), w.detach(), k, s, p, d)
    assert (y1 - y2).abs().max() < 1e-9
    gout = torch.full_like(y2, 1.0 / y2.numel())  # d(mean)/dy, as in the reference's test (:254-260)
    gx2, gw2 = torch.autograd.grad(y2.mean(), (x, w))
    assert (cref.backward_input(gout, w.detach(), x.shape, k, s, p, d) - gx2).abs().max() < 1e-9
    assert (cref.backward_weight(gout, x.detach(), w.shape, k, s, p, d) - gw2).abs().max() < 1e-9


def test_integer_inputs_are_exact_and_padded_taps_zero():
    """index / pad bookkeeping: integer-valued inputs sum exactly; padded-tap weight grads are exactly 0"""
    g = torch.Generator().manual_seed(3)
    x = torch.randint(-4, 5, (2, 8, 6, 5), generator=g).float()
    w = torch.randint(-3, 4, (2, 1, 4, 9, 6, 5), generator=g).float()
    gout = torch.randint(-2, 3, (2, 8, 6, 5), generator=g).float()
    assert torch.equal(cref.forward(x, w, 3, 1, 1, 1), unfold_oracle.aggregation_unfold(x, w, 3, 1, 1, 1))
    gw = cref.backward_weight(gout, x, w.shape, 3, 1, 1, 1)
    assert torch.all(gw[:, :, :, 0, 0, :] == 0) and torch.all(gw[:, :, :, 0, :, 0] == 0)   # tap (0,0): top row / left col
    assert torch.all(gw[:, :, :, 8, -1, :] == 0) and torch.all(gw[:, :, :, 8, :, -1] == 0)  # tap (2,2): bottom / right


def test_out_size_matches_python_formula():
    for H in range(1, 20):
        for k in (1, 3, 5):
            for s in (1, 2, 3):
                for p in (0, 1, 2):
                    for d in (1, 2):
                        want = int((H + 2 * p - (d * (k - 1) + 1)) / s + 1)
                        assert cref.out_size(H, k, s, p, d) == want

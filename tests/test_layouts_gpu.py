"""The per-tensor-layout kernels (cot_*_lay) on the MI355X at the benchmark batch: bit-identical with the NCHW entry points
on the permuted tensors (tests/layout_cases.py).  The NCHW entry points themselves are compared element-wise with fp32 torch /
the oracle in tests/test_dispatch_parity_gpu.py and tests/test_step_kernels_b80_gpu.py."""
import ctypes

import pytest
import torch

from cotnet_amd import _lib
from tests import layout_cases as lc

pytestmark = pytest.mark.gpu
B = 80


def _st():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


@pytest.mark.parametrize("C,HW", [(256, 196), (128, 196), (1024, 196), (512, 49), (2048, 49)])
@pytest.mark.parametrize("act,res,y2,ps", [(1, False, True, False), (0, False, False, False), (2, False, False, False), (1, True, False, True)])
def test_batchnorm_forward_layouts(C, HW, act, res, y2, ps):
    lc.bn_forward_case(_lib.lib(), "cuda", _st(), B, C, HW, act, res, y2, ps)


@pytest.mark.parametrize("C,HW", [(256, 196), (128, 196), (1024, 196), (512, 49), (2048, 49)])
@pytest.mark.parametrize("act,res,dy2,ps", [(1, False, True, False), (0, False, False, False), (2, False, False, False), (1, True, False, True)])
def test_batchnorm_backward_layouts(C, HW, act, res, dy2, ps):
    lc.bn_backward_case(_lib.lib(), "cuda", _st(), B, C, HW, act, res, dy2, ps)


@pytest.mark.parametrize("C,HW", [(256, 196), (512, 49)])
def test_radix_tail_layouts(C, HW):
    lc.radix_case(_lib.lib(), "cuda", _st(), B, C, HW)


@pytest.mark.parametrize("G,HW", [(32, 196), (64, 49)])
def test_group_norm9_layouts(G, HW):
    lc.gn9_case(_lib.lib(), "cuda", _st(), B, G, HW)


@pytest.mark.parametrize("Ci,Co,HW,act,res,mask", [(256, 64, 3136, 1, False, False), (64, 256, 3136, 1, True, True), (128, 32, 3136, 1, False, False),
                                                   (64, 64, 3136, 0, False, False), (128, 512, 784, 1, True, True)])
def test_batchnorm_statistics_from_the_convolution_epilogue(Ci, Co, HW, act, res, mask):
    """the conv1x1 -> BatchNorm pairs of the 56 x 56 stage (and a 28 x 28 one) at the benchmark batch"""
    lc.bn_epilogue_case(_lib.lib(), "cuda", _st(), B, Ci, Co, HW, act, res, mask)


@pytest.mark.parametrize("N,Ci,Co,HW,split", [(1, 1024, 256, 15680, 0), (1, 256, 1024, 15680, 0), (1, 512, 128, 15680, 256), (1, 2048, 512, 3920, 0),
                                              (1, 512, 2048, 3920, 0), (80, 512, 128, 784, 0), (80, 64, 256, 3136, 0), (80, 256, 64, 3136, 0),
                                              (1, 128, 288, 15680, 0)])
def test_conv1x1_lds_layouts_do_not_change_the_result(N, Ci, Co, HW, split, request):
    """cot_set_tuning(48): the bank-conflict-free LDS layouts of the 1x1 forward / data-gradient kernel (X stage of the 128-pixel tiles
    permuted per channel row; W and transposed-W chunk permutations) against the layouts of rounds 2-4 on the MI355X --
    the same bits, on the channel-major deep layers and on NCHW layers of the benchmark step"""
    L, st, BF = _lib.lib(), _st(), _lib.COT_BF16
    request.addfinalizer(lambda: L.cot_set_tuning(48, 7))
    torch.manual_seed(5)
    P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None  # noqa: E731
    x1 = torch.randn(N, split or Ci, HW, device="cuda").bfloat16()
    x2 = torch.randn(N, Ci - split, HW, device="cuda").bfloat16() if split else None
    w = (torch.randn(Co, Ci, device="cuda") / Ci ** 0.5).bfloat16()
    gy = torch.randn(N, Co, HW, device="cuda").bfloat16()
    ws = torch.empty(int(L.cot_conv1x1_workspace(N, Ci, Co, HW, 0)), dtype=torch.uint8, device="cuda")
    outs = []
    for key in (7, 0):
        assert L.cot_set_tuning(48, key) == 0
        y = torch.full((N, Co, HW), float("nan"), device="cuda").bfloat16()
        g1 = torch.full_like(x1, float("nan"))
        g2 = torch.full_like(x2, float("nan")) if split else None
        assert L.cot_conv1x1_forward(P(x1), P(x2), split or Ci, P(w), None, P(y), N, Ci, Co, HW, BF, st) == 0, L.cot_last_error()
        assert L.cot_conv1x1_backward_data(P(gy), P(w), P(g1), P(g2), split or Ci, 0, P(ws), N, Ci, Co, HW, BF, st) == 0, L.cot_last_error()
        torch.cuda.synchronize()
        outs.append((y, g1, g2))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    if split:
        assert torch.equal(outs[0][2], outs[1][2])
    assert not torch.isnan(outs[0][0].float()).any() and not torch.isnan(outs[0][1].float()).any()
    xf = torch.cat([x1, x2], 1).float() if split else x1.float()
    ref = torch.matmul(w.float(), xf[0])
    assert torch.allclose(outs[0][0][0].float(), ref, atol=3e-2, rtol=2e-2)

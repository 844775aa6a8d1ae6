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

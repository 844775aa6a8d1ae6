"""BatchNorm + SiLU of the aggregation's output folded into the radix tail (cot_bn_batch_stats + cot_radix_*_bn; reference
models/cotnet.py:89-104) on the MI355X, through the C ABI: forward against the unfused kernels (bit for bit where those ran the
streaming BatchNorm), both directions against torch autograd of the reference formula; CoTNet-50's four stage shapes at the recipe
batch and small / ragged ones, NCHW and the deep stages' channel-major k / output."""
import pytest
import torch

from cotnet_amd import _lib
from tests.bn_tail_cases import bn_tail_case, relu_res_case, rowstats_case

pytestmark = pytest.mark.gpu

STEP = [(80, 64, 56, 56), (80, 128, 28, 28), (80, 256, 14, 14), (80, 512, 7, 7)]
SMALL = [(3, 8, 8, 8), (5, 12, 7, 7), (2, 4, 3, 5), (9, 3, 7, 7), (4, 16, 14, 14), (2, 8, 24, 24), (33, 5, 7, 7)]


def _lib_on_device():
    L = _lib.lib()
    L._test_device = "cuda"
    return L


@pytest.mark.parametrize("sums", [0, 1])
@pytest.mark.parametrize("lay_k", [0, 1])
@pytest.mark.parametrize("shape", STEP + SMALL)
def test_bn_tail_bf16(shape, lay_k, sums):
    bn_tail_case(_lib_on_device(), *shape, torch.bfloat16, lay_k, bool(sums))
    torch.cuda.synchronize()


@pytest.mark.parametrize("sums", [0, 1])
@pytest.mark.parametrize("lay_k", [0, 1])
@pytest.mark.parametrize("shape", SMALL + [(8, 64, 56, 56)])
def test_bn_tail_fp32(shape, lay_k, sums):
    bn_tail_case(_lib_on_device(), *shape, torch.float32, lay_k, bool(sums))
    torch.cuda.synchronize()


@pytest.mark.parametrize("gn", [0, 1])
@pytest.mark.parametrize("shape", STEP + [(2, 64, 8, 8), (3, 64, 7, 7), (1, 128, 5, 8), (3, 192, 7, 7), (5, 64, 20, 20)])
def test_agg_forward_rowstats(shape, gn):
    """cot_agg_forward_rowstats + cot_bn_rowstats_finalize: output bit-identical to the plain forward, row sums and statistics against torch"""
    if gn and shape[3] % 2:
        pytest.skip("the GroupNorm prologue takes even rows (cot_agg_gn9_forward)")
    rowstats_case(_lib_on_device(), *shape, gn)
    torch.cuda.synchronize()


@pytest.mark.parametrize("N,Ci,Co,HW", [(80, 256, 64, 3136), (80, 512, 128, 784), (1, 1024, 256, 15680), (1, 2048, 512, 3920),
                                        (2, 64, 32, 784), (5, 128, 64, 64), (20, 64, 32, 16), (1, 96, 64, 392)])  # (N * HW > 256: the sign mask's geometries)
def test_conv1x1_data_gradient_with_the_masked_residual(N, Ci, Co, HW):
    """cot_conv1x1_backward_data_relu_res on CoTNet-50's conv1 shapes at B = 80 (NCHW and channel-major) and small ones: bit-identical to
    the residual gradient materialised + cot_conv1x1_backward_data(accumulate = 1)"""
    relu_res_case(_lib_on_device(), N, Ci, Co, HW)
    torch.cuda.synchronize()

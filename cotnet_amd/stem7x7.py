"""The backbone's first convolution (7x7, stride 2, padding 3, 3 -> 64) on csrc/stem7x7.hip (opt-in COT_STEM=hip).

`stem_conv(conv, x)` evaluates the default stem's `conv1 = nn.Conv2d(3, 64, 7, stride=2, padding=3, bias=False)`
(models/resnet.py:539-555) -- same parameter, same state_dict -- with the library's MFMA implicit-GEMM kernels (forward and
a deterministic weight gradient; the network input takes no gradient).  With the blocks, poolings and the head on the
library as well (bench.py's `new` kernel set) no MIOpen / BLAS call is left in a CoTNet-50 training step.
Eligible: bf16 (MFMA implicit GEMM) or fp32 (plain fp32 kernels, csrc/stem7x7_f32.hip: the reference's own precision) NCHW-contiguous
input that does not require grad, output width a multiple of 8 (224 / 256 / 288 / 320
inputs); anything else takes the module.
"""
import ctypes
import os

import torch
from torch import nn
from torch.autograd import Function

from . import _lib, grad_sink

MODE = os.environ.get("COT_STEM", "hip")  # default: the library's kernels; COT_STEM=module opts out
_DEVICE_ONLY = True  # tests drive the autograd wiring on CPU tensors through the host-emulated kernels
_WS = _lib.register_cache({})


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream) if _DEVICE_ONLY else None


def _ws_bytes(N, H, W):
    k = (N, H, W)
    v = _WS.get(k)
    if v is None:
        v = _WS[k] = int(_lib.lib().cot_stem7x7s2_workspace(N, H, W))
    return v


class _Stem(Function):
    @staticmethod
    def forward(ctx, x, weight):
        N, _, H, W = x.shape
        y = torch.empty((N, 64, (H - 1) // 2 + 1, (W - 1) // 2 + 1), dtype=x.dtype, device=x.device)
        rc = _lib.lib().cot_stem7x7s2_forward(x.data_ptr(), weight.data_ptr(), y.data_ptr(), N, H, W, _lib.dtype_code(x.dtype),
                                              _stream())
        if rc:
            _lib.check(rc, "cot_stem7x7s2_forward")
        ctx.save_for_backward(x, weight)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, weight = ctx.saved_tensors
        N, _, H, W = x.shape
        gy = gy.contiguous()
        ws = torch.empty(_ws_bytes(N, H, W), dtype=torch.uint8, device=gy.device)
        gw = grad_sink.out_like(weight)
        rc = _lib.lib().cot_stem7x7s2_backward_weight(gy.data_ptr(), x.data_ptr(), gw.data_ptr(), ws.data_ptr(), N, H, W,
                                                      _lib.dtype_code(x.dtype), _stream())
        if rc:
            _lib.check(rc, "cot_stem7x7s2_backward_weight")
        return None, gw


def eligible(conv, x):
    return (MODE == "hip" and isinstance(conv, nn.Conv2d) and conv.in_channels == 3 and conv.out_channels == 64
            and conv.kernel_size == (7, 7) and conv.stride == (2, 2) and conv.padding == (3, 3) and conv.dilation == (1, 1)
            and conv.groups == 1 and conv.bias is None and (x.is_cuda or not _DEVICE_ONLY) and x.dim() == 4
            and x.shape[1] == 3 and x.dtype in (torch.bfloat16, torch.float32) and conv.weight.dtype == x.dtype
            and x.is_contiguous() and conv.weight.is_contiguous() and not x.requires_grad
            and x.data_ptr() % 16 == 0 and _ws_bytes(x.shape[0], x.shape[2], x.shape[3]) > 0)


def stem_conv(conv, x):
    """`conv(x)`; see the module docstring for when the library kernels serve it"""
    if MODE == "hip" and eligible(conv, x):
        return _Stem.apply(x, conv.weight)
    if MODE == "hip":
        _lib.fallback("stem_conv", x, f"-> {conv.out_channels}, kernel {tuple(conv.kernel_size)}")
    return conv(x)

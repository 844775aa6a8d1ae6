"""A deep stem's first convolution (3x3, stride 2, padding 1, 3 -> 32 / 64) on csrc/stem3x3.hip (COT_STEM=hip, the default).

`stem3x3_conv(conv, x)` evaluates `nn.Conv2d(in_chans, stem_chs_1, 3, stride=2, padding=1, bias=False)` of a 'deep' stem
(models/cotnet_hybrid.py:359, the SE-CoTNetD models; models/resnet.py deep stems) -- same parameter, same state_dict -- with the
library's MFMA implicit-GEMM kernels (forward and a deterministic weight gradient; the network input takes no gradient).  The
stem's two stride-1 3x3 convolutions go through cotnet_amd.conv3x3g (groups = 1): with both, an SE-CoTNetD training step holds no
vendor convolution.  Eligible: bf16 NCHW-contiguous input that does not require grad, 32 or 64 output channels, output width a
multiple of 8; anything else takes the module (counted by _lib.fallback).
"""
import ctypes

import torch
from torch import nn
from torch.autograd import Function

from . import _lib, grad_sink, stem7x7

_WS = _lib.register_cache({})


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream) if stem7x7._DEVICE_ONLY else None


def _ws_bytes(N, H, W, Co):
    k = (N, H, W, Co)
    v = _WS.get(k)
    if v is None:
        v = _WS[k] = int(_lib.lib().cot_stem3x3s2_workspace(N, H, W, Co))
    return v


class _Stem3x3(Function):
    @staticmethod
    def forward(ctx, x, weight):
        N, _, H, W = x.shape
        Co = weight.shape[0]
        y = torch.empty((N, Co, (H - 1) // 2 + 1, (W - 1) // 2 + 1), dtype=x.dtype, device=x.device)
        rc = _lib.lib().cot_stem3x3s2_forward(x.data_ptr(), weight.data_ptr(), y.data_ptr(), N, H, W, Co, _lib.COT_BF16, _stream())
        if rc:
            _lib.check(rc, "cot_stem3x3s2_forward")
        ctx.save_for_backward(x, weight)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, weight = ctx.saved_tensors
        N, _, H, W = x.shape
        Co = weight.shape[0]
        gy = gy.contiguous()
        ws = torch.empty(_ws_bytes(N, H, W, Co), dtype=torch.uint8, device=gy.device)
        gw = grad_sink.out_like(weight)
        rc = _lib.lib().cot_stem3x3s2_backward_weight(gy.data_ptr(), x.data_ptr(), gw.data_ptr(), ws.data_ptr(), N, H, W, Co,
                                                      _lib.COT_BF16, _stream())
        if rc:
            _lib.check(rc, "cot_stem3x3s2_backward_weight")
        return None, gw


def eligible(conv, x):
    return (stem7x7.MODE == "hip" and isinstance(conv, nn.Conv2d) and conv.in_channels == 3 and conv.out_channels in (32, 64)
            and conv.kernel_size == (3, 3) and conv.stride == (2, 2) and conv.padding == (1, 1) and conv.dilation == (1, 1)
            and conv.groups == 1 and conv.bias is None and (x.is_cuda or not stem7x7._DEVICE_ONLY) and x.dim() == 4
            and x.shape[1] == 3 and x.dtype == torch.bfloat16 and conv.weight.dtype == torch.bfloat16
            and x.is_contiguous() and conv.weight.is_contiguous() and not x.requires_grad
            and x.data_ptr() % 16 == 0 and _ws_bytes(x.shape[0], x.shape[2], x.shape[3], conv.out_channels) > 0)


def stem3x3_conv(conv, x):
    """`conv(x)`; see the module docstring for when the library kernels serve it"""
    if stem7x7.MODE == "hip" and eligible(conv, x):
        return _Stem3x3.apply(x, conv.weight)
    if stem7x7.MODE == "hip":
        _lib.fallback("deep_stem_conv", x, f"-> {conv.out_channels}, kernel {tuple(conv.kernel_size)}, stride {tuple(conv.stride)}")
    return conv(x)

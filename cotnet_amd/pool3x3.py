"""The backbone's two 3x3 / stride-2 / padding-1 poolings on csrc/pool3x3.hip (opt-in COT_POOL=hip).

`pool(module, x)` evaluates `nn.MaxPool2d(3, 2, 1)` (after the stem, models/resnet.py:556-561) or `nn.AvgPool2d(3, 2,
padding=1)` (the "avd" pooling of stride-2 bottlenecks, models/cotnet.py:216) with kernels whose backward runs at
memory speed -- torch's max_pool_backward / avg_pool2d_backward were 436 us and 3 x 163 us of the round-1 step for ~30 us of
traffic each; the max-pool forward keeps the arg-max as one byte per window (torch's tie rule) and the backward reads that instead
of an int64 index tensor (or x).
Also `nn.AvgPool2d(2, 2)` on even planes (the pooling of an `avg_down` shortcut, models/resnet.py:380-394).
Any other module or tensor (other geometry, ceil_mode, fp64, channels-last, CPU) takes the module itself.
"""
import ctypes
import os

import torch
from torch import nn
from torch.autograd import Function

from . import _lib

MODE = os.environ.get("COT_POOL", "hip")  # default: the library's kernels; COT_POOL=module opts out
_DEVICE_ONLY = True  # tests drive the autograd wiring on CPU tensors through the host-emulated kernels
_DT = {torch.float32: _lib.COT_F32, torch.bfloat16: _lib.COT_BF16}


def _p(t):
    return ctypes.c_void_p(t.data_ptr())


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream) if _DEVICE_ONLY else None


def _out(x):
    N, C, H, W = x.shape
    return torch.empty((N, C, (H - 1) // 2 + 1, (W - 1) // 2 + 1), dtype=x.dtype, device=x.device)


class _AvgPool(Function):
    @staticmethod
    def forward(ctx, x):
        N, C, H, W = x.shape
        y = _out(x)
        rc = _lib.lib().cot_avgpool3x3s2_forward(_p(x), _p(y), N * C, H, W, _DT[x.dtype], _stream())
        if rc:
            _lib.check(rc, "cot_avgpool3x3s2_forward")
        ctx.shape = x.shape
        return y

    @staticmethod
    def backward(ctx, gy):
        N, C, H, W = ctx.shape
        gy = gy.contiguous()
        gx = torch.empty(ctx.shape, dtype=gy.dtype, device=gy.device)
        rc = _lib.lib().cot_avgpool3x3s2_backward(_p(gy), _p(gx), N * C, H, W, _DT[gy.dtype], _stream())
        if rc:
            _lib.check(rc, "cot_avgpool3x3s2_backward")
        return gx


class _AvgPool2(Function):
    """nn.AvgPool2d(2, 2) on even planes (downsample_avg's pooling, models/resnet.py:380-394)"""
    @staticmethod
    def forward(ctx, x):
        N, C, H, W = x.shape
        y = torch.empty((N, C, H // 2, W // 2), dtype=x.dtype, device=x.device)
        rc = _lib.lib().cot_avgpool2x2s2_forward(_p(x), _p(y), N * C, H, W, _DT[x.dtype], _stream())
        if rc:
            _lib.check(rc, "cot_avgpool2x2s2_forward")
        ctx.shape = x.shape
        return y

    @staticmethod
    def backward(ctx, gy):
        N, C, H, W = ctx.shape
        gy = gy.contiguous()
        gx = torch.empty(ctx.shape, dtype=gy.dtype, device=gy.device)
        rc = _lib.lib().cot_avgpool2x2s2_backward(_p(gy), _p(gx), N * C, H, W, _DT[gy.dtype], _stream())
        if rc:
            _lib.check(rc, "cot_avgpool2x2s2_backward")
        return gx


class _MaxPool(Function):
    @staticmethod
    def forward(ctx, x):
        N, C, H, W = x.shape
        y = _out(x)
        taps = torch.empty(y.shape, dtype=torch.uint8, device=x.device)  # winning element of every window, one byte
        rc = _lib.lib().cot_maxpool3x3s2_forward_taps(_p(x), _p(y), _p(taps), N * C, H, W, _DT[x.dtype], _stream())
        if rc:
            _lib.check(rc, "cot_maxpool3x3s2_forward_taps")
        ctx.save_for_backward(taps)
        ctx.shape = x.shape
        return y

    @staticmethod
    def backward(ctx, gy):
        (taps,) = ctx.saved_tensors
        N, C, H, W = ctx.shape
        gy = gy.contiguous()
        gx = torch.empty(ctx.shape, dtype=gy.dtype, device=gy.device)
        rc = _lib.lib().cot_maxpool3x3s2_backward_taps(_p(gy), _p(taps), _p(gx), N * C, H, W, _DT[gy.dtype], _stream())
        if rc:
            _lib.check(rc, "cot_maxpool3x3s2_backward_taps")
        return gx


class _BlurPool(Function):
    """BlurPool2d(filt_size=3, stride=2): reflection pad 1 + depthwise [1 2 1] x [1 2 1] / 16, stride 2 (blur_pool.py:53-58)"""

    @staticmethod
    def forward(ctx, x):
        N, C, H, W = x.shape
        y = _out(x)
        rc = _lib.lib().cot_blurpool3x3s2_forward(_p(x), _p(y), N * C, H, W, _DT[x.dtype], _stream())
        if rc:
            _lib.check(rc, "cot_blurpool3x3s2_forward")
        ctx.shape = x.shape
        return y

    @staticmethod
    def backward(ctx, gy):
        N, C, H, W = ctx.shape
        gy = gy.contiguous()
        gx = torch.empty(ctx.shape, dtype=gy.dtype, device=gy.device)
        rc = _lib.lib().cot_blurpool3x3s2_backward(_p(gy), _p(gx), N * C, H, W, _DT[gy.dtype], _stream())
        if rc:
            _lib.check(rc, "cot_blurpool3x3s2_backward")
        return gx


def blur_eligible(x):
    return ((x.is_cuda or not _DEVICE_ONLY) and x.dim() == 4 and x.dtype in _DT and x.is_contiguous() and x.shape[2] >= 2
            and x.shape[3] >= 2)


def blur_pool(x):
    return _BlurPool.apply(x)


def _pair(v):
    return (v, v) if isinstance(v, int) else tuple(v)


def eligible(module, x):
    if not (MODE == "hip" and (x.is_cuda or not _DEVICE_ONLY) and x.dim() == 4 and x.dtype in _DT and x.is_contiguous()):
        return False
    if isinstance(module, nn.MaxPool2d):
        return (_pair(module.kernel_size) == (3, 3) and _pair(module.stride) == (2, 2) and _pair(module.padding) == (1, 1)
                and _pair(module.dilation) == (1, 1) and not module.ceil_mode and not module.return_indices)
    if isinstance(module, nn.AvgPool2d) and _pair(module.kernel_size) == (2, 2):
        # even planes: every window is a full 2 x 2 block whatever ceil_mode / count_include_pad say
        return (_pair(module.stride) == (2, 2) and _pair(module.padding) == (0, 0) and module.divisor_override is None
                and x.shape[2] % 2 == 0 and x.shape[3] % 2 == 0)
    if isinstance(module, nn.AvgPool2d):
        return (_pair(module.kernel_size) == (3, 3) and _pair(module.stride) == (2, 2) and _pair(module.padding) == (1, 1)
                and not module.ceil_mode and module.count_include_pad and module.divisor_override is None)
    return False


def pool(module, x):
    """`module(x)`; the two backbone poolings go through the HIP kernels when COT_POOL=hip and the tensor qualifies"""
    if MODE == "hip" and eligible(module, x):
        if isinstance(module, nn.MaxPool2d):
            return _MaxPool.apply(x)
        return (_AvgPool2 if _pair(module.kernel_size) == (2, 2) else _AvgPool).apply(x)
    if MODE == "hip":
        _lib.fallback("pool", x, type(module).__name__)
    return module(x)

"""1x1 convolution on NCHW tensors without layout changes.

Round-1 profile: MIOpen runs every convolution through NHWC kernels and brackets each call with
NCHW<->NHWC transposes and cast/zero kernels -- 6.7 + 2.3 ms of a 36 ms step, more than the GEMMs themselves for the
1x1 convolutions.  In NCHW a 1x1 convolution is, per image, Y[n] = W[Co,Ci] @ X[n][Ci,HW]; forward and the data
gradient (gX[n] = W^T @ gY[n]) need no transposes at all, the weight gradient reduces over the contiguous pixel index.

`conv1x1(conv, x, x2=None)` evaluates an ordinary `nn.Conv2d` (same parameters / state_dict as the reference's
modules: CotLayer.embed[0], embed[3], conv1x1[0] -- models/cotnet.py:51-62 -- Bottleneck.conv1 / conv3 / downsample)
on `x`, or on the channel concatenation `[x, x2]` WITHOUT materialising it (the reference's `torch.cat([x, k], dim=1)`,
models/cotnet.py:81).  Implementation is chosen by COT_CONV1X1:

    (unset)  the module itself (MIOpen)                                   -- the library default; bench.py's `new` set uses hip
    hip      hand-written MFMA kernels behind cot_conv1x1_* / cot_conv1x1g_*:
               bf16, groups 1, Ci % 8 == 0, Co % 8 == 0 -> csrc/conv_lds.hip (LDS-DMA pipeline) / csrc/conv1x1.hip
               fp32, grouped (CoXtLayer, groups 2), other channel counts -> csrc/conv_gen.hip (general kernels)
    matmul   torch.matmul / einsum batched GEMMs (rocBLAS / hipBLASLt)

`hip` is verified against torch through the host emulation of the kernels (tests/test_kernels_emulated.py) and on the MI355X
by tests/test_conv1x1_gpu.py / tests/test_conv_general_gpu.py.
"""
import ctypes
import os

import torch
from torch.autograd import Function

from . import _lib

MODE = os.environ.get("COT_CONV1X1", "hip")  # default: the library's kernels; COT_CONV1X1=module opts out
_DEVICE_ONLY = True  # tests drive the autograd wiring on CPU tensors through the host-emulated kernels


class _Conv1x1(Function):
    @staticmethod
    def forward(ctx, x, weight, bias):
        N, Ci, H, W = x.shape
        Co = weight.shape[0]
        w2 = weight.view(Co, Ci)
        y = torch.matmul(w2, x.view(N, Ci, H * W))
        if bias is not None:
            y = y + bias.view(1, Co, 1)
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        return y.view(N, Co, H, W)

    @staticmethod
    def backward(ctx, gy):
        x, weight = ctx.saved_tensors
        N, Ci, H, W = x.shape
        Co = weight.shape[0]
        gy3 = gy.contiguous().view(N, Co, H * W)
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            gx = torch.matmul(weight.view(Co, Ci).t(), gy3).view(N, Ci, H, W)
        if ctx.needs_input_grad[1]:
            gw = torch.einsum("nop,nip->oi", gy3, x.view(N, Ci, H * W)).view_as(weight).to(weight.dtype)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            gb = gy3.sum((0, 2)).to(weight.dtype)
        return gx, gw, gb


def _p(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream) if _DEVICE_ONLY else None


_WS = _lib.register_cache({})  # (N, Ci, Co, HW, has_bias) -> backward workspace bytes (pure function of the shape)


def _ws_bytes(N, Ci, Co, HW, has_bias):
    k = (N, Ci, Co, HW, has_bias)
    v = _WS.get(k)
    if v is None:
        v = _WS[k] = int(_lib.lib().cot_conv1x1_workspace(N, Ci, Co, HW, has_bias))
    return v


class _Conv1x1Hip(Function):
    """y = conv1x1([x1 | x2], weight, bias) through libcotnet_hip.so; x2 may be None"""

    @staticmethod
    def forward(ctx, x1, x2, weight, bias):
        N, c1, H, W = x1.shape
        Co, Ci = weight.shape[0], weight.shape[1]
        y = torch.empty((N, Co, H, W), dtype=x1.dtype, device=x1.device)
        rc = _lib.lib().cot_conv1x1_forward(_p(x1), _p(x2), c1, _p(weight), _p(bias), _p(y), N, Ci, Co, H * W,
                                            _lib.dtype_code(x1.dtype), _stream())
        if rc:
            _lib.check(rc, "cot_conv1x1_forward")
        ctx.save_for_backward(x1, x2, weight)
        ctx.has_bias = bias is not None
        return y

    @staticmethod
    def backward(ctx, gy):
        x1, x2, weight = ctx.saved_tensors
        N, c1, H, W = x1.shape
        Co, Ci = weight.shape[0], weight.shape[1]
        HW = H * W
        gy = gy.contiguous()
        L = _lib.lib()
        has_bias = 1 if ctx.has_bias else 0
        dt = _lib.dtype_code(x1.dtype)
        nbytes = (int(L.cot_convg_workspace(N, Ci, Co, 1, HW, 1, 1)) if x1.dtype == torch.float32
                  else _ws_bytes(N, Ci, Co, HW, has_bias))  # fp32: the general kernels' partial sums
        ws = torch.empty(nbytes, dtype=torch.uint8, device=gy.device)
        gx1 = gx2 = gw = gb = None
        if ctx.needs_input_grad[0] or (x2 is not None and ctx.needs_input_grad[1]):
            gx1 = torch.empty_like(x1)
            gx2 = torch.empty_like(x2) if x2 is not None else None
            rc = L.cot_conv1x1_backward_data(_p(gy), _p(weight), _p(gx1), _p(gx2), c1, 0, _p(ws), N, Ci, Co, HW,
                                             dt, _stream())
            if rc:
                _lib.check(rc, "cot_conv1x1_backward_data")
        if ctx.needs_input_grad[2] or (ctx.has_bias and ctx.needs_input_grad[3]):
            gw = torch.empty_like(weight)
            gb = torch.empty(Co, dtype=weight.dtype, device=gy.device) if ctx.has_bias else None
            rc = L.cot_conv1x1_backward_weight(_p(gy), _p(x1), _p(x2), c1, _p(gw), _p(gb), _p(ws), N, Ci, Co, HW,
                                               dt, _stream())
            if rc:
                _lib.check(rc, "cot_conv1x1_backward_weight")
        return gx1, gx2, gw, gb


class _Conv1x1gHip(Function):
    """y = conv1x1(x, weight, bias, groups) through the general kernels (cot_conv1x1g_*, csrc/conv_gen.hip): grouped 1x1
    convolutions of CoXtLayer (models/cotnet.py:123-131) and channel counts the tuned bf16 kernels do not tile"""

    @staticmethod
    def forward(ctx, x, weight, bias, groups):
        N, Ci, H, W = x.shape
        Co = weight.shape[0]
        y = torch.empty((N, Co, H, W), dtype=x.dtype, device=x.device)
        rc = _lib.lib().cot_conv1x1g_forward(_p(x), _p(weight), _p(bias), _p(y), N, Ci, Co, groups, H * W,
                                             _lib.dtype_code(x.dtype), _stream())
        if rc:
            _lib.check(rc, "cot_conv1x1g_forward")
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        ctx.groups = groups
        return y

    @staticmethod
    def backward(ctx, gy):
        x, weight = ctx.saved_tensors
        N, Ci, H, W = x.shape
        Co, G, HW = weight.shape[0], ctx.groups, H * W
        gy = gy.contiguous()
        L = _lib.lib()
        dt = _lib.dtype_code(x.dtype)
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            gx = torch.empty_like(x)
            rc = L.cot_conv1x1g_backward_data(_p(gy), _p(weight), _p(gx), 0, N, Ci, Co, G, HW, dt, _stream())
            if rc:
                _lib.check(rc, "cot_conv1x1g_backward_data")
        if ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2]):
            ws = torch.empty(int(L.cot_convg_workspace(N, Ci, Co, G, HW, 1, 1)), dtype=torch.uint8, device=gy.device)
            gw = torch.empty_like(weight)
            gb = torch.empty(Co, dtype=weight.dtype, device=gy.device) if ctx.has_bias else None
            rc = L.cot_conv1x1g_backward_weight(_p(gy), _p(x), _p(gw), _p(gb), _p(ws), N, Ci, Co, G, HW, dt, _stream())
            if rc:
                _lib.check(rc, "cot_conv1x1g_backward_weight")
        return gx, gw, gb, None


def _is_1x1(conv):
    return (isinstance(conv, torch.nn.Conv2d) and conv.kernel_size == (1, 1) and conv.stride == (1, 1)
            and conv.padding == (0, 0) and conv.dilation == (1, 1))


def _plain_1x1(conv):
    return (isinstance(conv, torch.nn.Conv2d) and conv.kernel_size == (1, 1) and conv.stride == (1, 1)
            and conv.padding == (0, 0) and conv.groups == 1 and conv.dilation == (1, 1))


def eligible(conv, x):
    """matmul path"""
    return (MODE == "matmul" and _plain_1x1(conv) and x.dim() == 4 and x.is_contiguous()
            and x.dtype == conv.weight.dtype)


def eligible_general(conv, x):
    """the general kernels: any groups and channel counts, bf16 or fp32 (one input tensor)"""
    return (MODE == "hip" and _is_1x1(conv) and (x.is_cuda or not _DEVICE_ONLY) and x.dim() == 4
            and x.dtype in (torch.bfloat16, torch.float32) and conv.weight.dtype == x.dtype and x.is_contiguous()
            and conv.weight.is_contiguous() and x.shape[1] == conv.in_channels
            and (conv.bias is None or conv.bias.dtype == x.dtype))


def eligible_hip(conv, x, x2=None):
    if not (MODE == "hip" and _plain_1x1(conv) and (x.is_cuda or not _DEVICE_ONLY) and x.dim() == 4 and x.dtype == torch.bfloat16
            and conv.weight.dtype == torch.bfloat16 and x.is_contiguous() and conv.weight.is_contiguous()):
        return False
    if x2 is not None and not (x2.dtype == x.dtype and x2.is_contiguous() and x2.shape[0] == x.shape[0]
                               and x2.shape[2:] == x.shape[2:]):
        return False
    ci = x.shape[1] + (x2.shape[1] if x2 is not None else 0)
    return ci == conv.in_channels and ci % 8 == 0 and conv.out_channels % 8 == 0


def conv1x1(conv, x, x2=None):
    """`conv(x)`, or `conv(torch.cat([x, x2], 1))` when x2 is given, for an nn.Conv2d (see the module docstring)."""
    if MODE == "hip" and eligible_hip(conv, x, x2):
        return _Conv1x1Hip.apply(x, x2, conv.weight, conv.bias)
    if x2 is not None:
        x = torch.cat([x, x2], dim=1)
    if MODE == "hip" and eligible_general(conv, x):
        if conv.groups == 1 and x.dtype == torch.float32:
            return _Conv1x1Hip.apply(x, None, conv.weight, conv.bias)
        return _Conv1x1gHip.apply(x, conv.weight, conv.bias, conv.groups)
    if MODE == "matmul" and eligible(conv, x):
        return _Conv1x1.apply(x, conv.weight, conv.bias)
    if MODE == "hip":
        _lib.fallback("conv1x1", x, f"-> {conv.out_channels}, kernel {tuple(conv.kernel_size)}, stride {tuple(conv.stride)}, groups {conv.groups}")
    return conv(x)


def run_downsample(ds, x):
    """`ds(x)` for the residual branch's nn.Sequential ([pool,] 1x1 conv, norm -- models/resnet.py:364-394) with the
    convolution routed through `conv1x1`.  A stride-2 1x1 convolution (the stage-entry projection when avg_down is off)
    is the stride-1 one on every second pixel: subsample, then the same kernels; its BatchNorm goes through the fused
    BatchNorm kernels instead of MIOpen."""
    if MODE and isinstance(ds, torch.nn.Sequential):
        from .fused_bn import fused_bn_act
        from .pool3x3 import pool
        for m in ds:
            if isinstance(m, torch.nn.Conv2d):
                if (MODE == "hip" and m.kernel_size == (1, 1) and m.stride == (2, 2) and m.padding == (0, 0)
                        and m.groups == 1 and x.dim() == 4 and x.dtype in (torch.bfloat16, torch.float32)
                        and m.weight.dtype == x.dtype and (x.is_cuda or not _DEVICE_ONLY)
                        and m.in_channels % 8 == 0 and m.out_channels % 8 == 0):  # (fp32: the general kernels, conv_gen.hip)
                    x = _Conv1x1Hip.apply(x[:, :, ::2, ::2].contiguous(), None, m.weight, m.bias)
                else:
                    x = conv1x1(m, x)
            elif isinstance(m, torch.nn.BatchNorm2d):
                x = fused_bn_act(x, m, None)  # (falls back to the module itself when not eligible)
            elif isinstance(m, torch.nn.AvgPool2d):
                x = pool(m, x)  # (downsample_avg's 2 x 2 pooling on the library's kernel when eligible, else the module)
            else:
                x = m(x)
        return x
    return ds(x)

"""Grouped 3x3 convolution (stride 1, padding 1) on NCHW bf16 tensors without layout changes.

`conv3x3(conv, x)` evaluates an ordinary `nn.Conv2d(kernel_size=3, stride=1, padding=1, bias=False)` -- the reference's
`CotLayer.key_embed[0]` (groups 4, models/cotnet.py:43-47) and `CoXtLayer.key_embed[0]` (groups 8, :112-116); same
parameter, same state_dict -- through the hand-written MFMA kernels of csrc/conv3x3g.hip (cot_conv3x3g_*) when
COT_CONV3X3=hip and the tensor qualifies (bf16 or fp32, NCHW-contiguous); otherwise through the module itself (MIOpen).
bf16 with channels per group a multiple of 8 runs the tuned kernels (csrc/conv_lds.hip, conv3x3g.hip), fp32 and other
channel counts (CoXtLayer: 12 / 24 per group) the general ones (csrc/conv_gen.hip).  Verified against torch through the
host emulation of the kernels (tests/test_kernels_emulated.py) and on the device by tests/test_conv3x3g_gpu.py.
"""
import ctypes
import os

import torch
from torch.autograd import Function

from . import _lib

MODE = os.environ.get("COT_CONV3X3", "hip")  # default: the library's kernels; COT_CONV3X3=module opts out
_DEVICE_ONLY = True  # tests drive the autograd wiring on CPU tensors through the host-emulated kernels

_MASKS = {}  # (H, W, device) -> uint8 tensor holding the per-pixel tap-validity table (read-only after creation)
_WS = _lib.register_cache({})     # (N, Cin, Cout, G, H, W) -> workspace bytes


def _p(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream) if _DEVICE_ONLY else None


def _masks(H, W, device):
    k = (H, W, str(device))
    m = _MASKS.get(k)
    if m is None:
        L = _lib.lib()
        m = torch.empty(int(L.cot_conv3x3g_masks_bytes(H, W)), dtype=torch.uint8, device=device)
        rc = L.cot_conv3x3g_masks(_p(m), H, W, _stream())
        if rc:
            _lib.check(rc, "cot_conv3x3g_masks")
        _MASKS[k] = m
    return m


def _ws_bytes(N, Cin, Cout, G, H, W, dtype=torch.bfloat16):
    """workspace bytes; fp32 tensors and channel counts off the tuned kernels' grid are served by the general kernels"""
    general = dtype == torch.float32 or (Cin // G) % 8 != 0 or (Cout // G) % 8 != 0
    key = (N, Cin, Cout, G, H, W, general)
    v = _WS.get(key)
    if v is None:
        L = _lib.lib()
        v = _WS[key] = max(int(L.cot_conv3x3g_workspace(N, Cin, Cout, G, H, W)),
                           int(L.cot_convg_workspace(N, Cin, Cout, G, H, W, 3)) if general else 0)
    return v


def new_guarded(N, C, H, W, dtype, dev):
    """[N, C, H, W] tensor with W + 1 (rounded up to 8) elements of the same allocation before and behind it: the LDS-staged weight
    gradient (cot_conv3x3g_backward_weight_guarded) copies x at pixel + tap offset in whole 16-byte pieces; what lies in the margins
    never reaches a sum.  Producers whose output feeds a 3x3 convolution allocate it this way (fused_bn_act)."""
    lead = (W + 1 + 7) // 8 * 8
    n = N * C * H * W
    flat = torch.empty(n + 2 * lead, dtype=dtype, device=dev)
    return flat[lead:lead + n].view(N, C, H, W)


def guard_elems(t):
    """elements of t's own allocation before its first and behind its last element (0 for a tensor that fills its storage)"""
    if not t.is_contiguous():
        return 0
    total = t.untyped_storage().nbytes() // t.element_size()
    return max(0, min(t.storage_offset(), total - t.storage_offset() - t.numel()))


class _Conv3x3G(Function):
    @staticmethod
    def forward(ctx, x, weight, groups):
        N, Cin, H, W = x.shape
        Cout = weight.shape[0]
        masks = _masks(H, W, x.device)
        ws = torch.empty(_ws_bytes(N, Cin, Cout, groups, H, W, x.dtype), dtype=torch.uint8, device=x.device)
        y = torch.empty((N, Cout, H, W), dtype=x.dtype, device=x.device)
        rc = _lib.lib().cot_conv3x3g_forward(_p(x), _p(weight), _p(y), _p(masks), _p(ws), N, Cin, Cout, groups, H, W,
                                             _lib.dtype_code(x.dtype), _stream())
        if rc:
            _lib.check(rc, "cot_conv3x3g_forward")
        ctx.save_for_backward(x, weight)
        ctx.groups = groups
        return y

    @staticmethod
    def backward(ctx, gy):
        x, weight = ctx.saved_tensors
        N, Cin, H, W = x.shape
        Cout, G = weight.shape[0], ctx.groups
        gy = gy.contiguous()
        L = _lib.lib()
        masks = _masks(H, W, x.device)
        ws = torch.empty(_ws_bytes(N, Cin, Cout, G, H, W, x.dtype), dtype=torch.uint8, device=x.device)
        gx = gw = None
        if ctx.needs_input_grad[0]:
            gx = torch.empty_like(x)
            rc = L.cot_conv3x3g_backward_data(_p(gy), _p(weight), _p(gx), 0, _p(masks), _p(ws), N, Cin, Cout, G, H, W,
                                              _lib.dtype_code(x.dtype), _stream())
            if rc:
                _lib.check(rc, "cot_conv3x3g_backward_data")
        if ctx.needs_input_grad[1]:
            gw = torch.empty_like(weight)
            # (the margins x's own allocation has around it: with W + 1 or more the LDS-staged kernel runs, with 0 the per-wave one)
            rc = L.cot_conv3x3g_backward_weight_guarded(_p(gy), _p(x), _p(gw), _p(masks), _p(ws), N, Cin, Cout, G, H, W,
                                                        _lib.dtype_code(x.dtype), guard_elems(x), _stream())
            if rc:
                _lib.check(rc, "cot_conv3x3g_backward_weight")
        return gx, gw, None


def eligible(conv, x):
    return (MODE == "hip" and isinstance(conv, torch.nn.Conv2d) and conv.kernel_size == (3, 3)
            and conv.stride == (1, 1) and conv.padding == (1, 1) and conv.dilation == (1, 1) and conv.bias is None
            and conv.padding_mode == "zeros" and (x.is_cuda or not _DEVICE_ONLY) and x.dim() == 4
            and x.dtype in (torch.bfloat16, torch.float32) and conv.weight.dtype == x.dtype and x.is_contiguous()
            and conv.weight.is_contiguous() and x.shape[1] == conv.in_channels)


def conv3x3(conv, x):
    """`conv(x)` for an nn.Conv2d; see the module docstring for when the HIP kernels serve it"""
    if MODE == "hip" and eligible(conv, x):
        return _Conv3x3G.apply(x, conv.weight, conv.groups)
    if MODE == "hip":
        _lib.fallback("conv3x3", x, f"-> {conv.out_channels}, stride {tuple(conv.stride)}, groups {conv.groups}")
    return conv(x)

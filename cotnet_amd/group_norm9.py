"""GroupNorm with 9 channels per group through csrc/group_norm9.hip, opt-in COT_GN9=hip.

`group_norm9(gn, x)` evaluates an ordinary `nn.GroupNorm(dim/8, 9*dim/8)` -- CotLayer.embed[4], models/cotnet.py:56 --
with one read and one write forward, two reads and one write backward (torch: 2R+1W and 6R+1W in 3 + 5 launches).
Eligible: bf16 (H*W <= 8192) or fp32 NCHW-contiguous tensors, affine parameters of the same dtype, 9 channels per group;
otherwise the module.
"""
import ctypes
import os

import torch
from torch.autograd import Function

from . import _lib

MODE = os.environ.get("COT_GN9", "hip")  # default: the library's kernel; COT_GN9=module opts out
_DEVICE_ONLY = True  # tests drive the autograd wiring on CPU tensors through the host-emulated kernels


def _p(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream) if _DEVICE_ONLY else None


class _GroupNorm9(Function):
    @staticmethod
    def forward(ctx, x, weight, bias, eps):
        N, C, H, W = x.shape
        G = C // 9
        y = torch.empty_like(x)
        stats = torch.empty(2 * N * G, dtype=torch.float32, device=x.device)
        rc = _lib.lib().cot_group_norm9_forward(_p(x), _p(weight), _p(bias), _p(y), _p(stats), _p(stats[N * G:]), N, C,
                                                H * W, eps, _lib.dtype_code(x.dtype), _stream())
        if rc:
            _lib.check(rc, "cot_group_norm9_forward")
        ctx.save_for_backward(x, weight, stats)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight, stats = ctx.saved_tensors
        N, C, H, W = x.shape
        G = C // 9
        dy = dy.contiguous()
        dx, dg, db = torch.empty_like(x), torch.empty_like(weight), torch.empty_like(weight)
        ws = torch.empty(2 * N * C, dtype=torch.float32, device=x.device)
        rc = _lib.lib().cot_group_norm9_backward(_p(dy), _p(x), _p(stats), _p(stats[N * G:]), _p(weight), _p(dx), _p(dg),
                                                 _p(db), _p(ws), N, C, H * W, _lib.dtype_code(x.dtype), _stream())
        if rc:
            _lib.check(rc, "cot_group_norm9_backward")
        return dx, dg, db, None


def eligible(gn, x):
    return (MODE == "hip" and isinstance(gn, torch.nn.GroupNorm) and gn.affine and gn.num_groups * 9 == gn.num_channels
            and (x.is_cuda or not _DEVICE_ONLY) and x.dim() == 4 and x.dtype in (torch.bfloat16, torch.float32)
            and x.is_contiguous() and x.shape[1] == gn.num_channels and gn.weight.dtype == x.dtype
            and (x.dtype == torch.float32 or x.shape[2] * x.shape[3] <= 8192) and x.data_ptr() % 16 == 0)


def group_norm9(gn, x):
    if MODE == "hip" and eligible(gn, x):
        return _GroupNorm9.apply(x, gn.weight, gn.bias, float(gn.eps))
    if MODE == "hip":
        _lib.fallback("group_norm9", x, f"groups {gn.num_groups}")
    return gn(x)

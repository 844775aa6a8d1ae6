"""Classifier head (global average pooling + fc) on the library's kernels (opt-in COT_HEAD=hip).

`head(pool, fc, x)` evaluates `fc(pool(x))` for the reference's `SelectAdaptivePool2d('avg', flatten=True)` +
`nn.Linear` head (models/resnet.py:570-574,:605-611): the pooled descriptor is written channel-major ([C][N]) by
`cot_radix_gap_t`, which makes the Linear layer a 1x1 convolution over one image of N pixels (`cot_conv1x1_*`, as for the
`se` branch) -- forward, data gradient and a deterministic weight gradient without a BLAS call.  Besides
the launch count, this keeps the whole step free of library kernels that accumulate into pre-zeroed buffers, which is what
HIP-graph replay tripped over in round 1.  Same parameters / state_dict as the modules it is applied to.
"""
import ctypes
import os

import torch
from torch import nn
from torch.autograd import Function

from . import _lib, grad_sink

MODE = os.environ.get("COT_HEAD", "hip")  # default: the library's kernels; COT_HEAD=module opts out
_DEVICE_ONLY = True  # tests drive the autograd wiring on CPU tensors through the host-emulated kernels
BF16 = _lib.COT_BF16
_WS = _lib.register_cache({})


def _p(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream) if _DEVICE_ONLY else None


def _ck(rc, what):
    if rc:
        _lib.check(rc, what)


class _Head(Function):
    @staticmethod
    def forward(ctx, x, weight, bias, drop_p=0.0):
        L = _lib.lib()
        N, C, H, W = x.shape
        O = weight.shape[0]
        st = _stream()
        gapT = torch.empty((C, N), dtype=x.dtype, device=x.device)
        dt = _lib.dtype_code(x.dtype)
        _ck(L.cot_radix_gap_t(_p(x), None, _p(gapT), N, C, H * W, dt, st), "cot_radix_gap_t")
        ctx.mask = None
        if drop_p > 0.0:  # F.dropout on the pooled descriptor (reference recipe: drop 0.25): C x N elements, two tiny launches
            # 0/1 keep mask, the 1/(1-p) scale applied in fp32 and the product rounded once -- as F.dropout does (a bf16 mask
            # with the scale baked in rounds 1.3333 to 1.3359: +0.2 % on every pooled feature in training)
            ctx.mask, ctx.scale = torch.rand((C, N), dtype=torch.float32, device=x.device) >= drop_p, 1.0 / (1.0 - drop_p)
            gapT = (gapT.float() * ctx.mask * ctx.scale).to(x.dtype)
        logT = torch.empty((O, N), dtype=x.dtype, device=x.device)
        _ck(L.cot_conv1x1_forward(_p(gapT), None, C, _p(weight), _p(bias), _p(logT), 1, C, O, N, dt, st),
            "cot_conv1x1_forward")
        ctx.save_for_backward(gapT, weight)
        ctx.shape, ctx.has_bias = x.shape, bias is not None
        return logT.t().contiguous()

    @staticmethod
    def backward(ctx, g):
        gapT, weight = ctx.saved_tensors
        L = _lib.lib()
        N, C, H, W = ctx.shape
        O = weight.shape[0]
        st = _stream()
        gT = g.t().contiguous()
        dt = _lib.dtype_code(gapT.dtype)
        key = (N, C, O, dt)
        nb = _WS.get(key)
        if nb is None:  # (fp32 runs on the general kernels, whose weight gradient has its own workspace)
            nb = _WS[key] = int(L.cot_convg_workspace(1, C, O, 1, N, 1, 1)) if dt == _lib.COT_F32 else int(L.cot_conv1x1_workspace(1, C, O, N, 1))
        ws = torch.empty(nb, dtype=torch.uint8, device=g.device)
        ggapT = torch.empty_like(gapT)
        _ck(L.cot_conv1x1_backward_data(_p(gT), _p(weight), _p(ggapT), None, C, 0, _p(ws), 1, C, O, N, dt, st),
            "cot_conv1x1_backward_data")
        gw = grad_sink.out_like(weight)
        gb = torch.empty(O, dtype=weight.dtype, device=g.device) if ctx.has_bias else None
        _ck(L.cot_conv1x1_backward_weight(_p(gT), _p(gapT), None, C, _p(gw), _p(gb), _p(ws), 1, C, O, N, dt, st),
            "cot_conv1x1_backward_weight")
        gf = ggapT.t().float()
        if ctx.mask is not None:
            gf = gf * (ctx.mask.t() * ctx.scale)
        gx = (gf / (H * W)).to(g.dtype).reshape(N, C, 1, 1).expand(N, C, H, W)  # d mean_hw
        return gx, gw, gb, None


def eligible(pool, fc, x):
    return (MODE == "hip" and isinstance(fc, nn.Linear) and getattr(pool, "pool_type", None) == "avg"
            and getattr(pool, "flatten", False) and (x.is_cuda or not _DEVICE_ONLY) and x.dim() == 4
            and x.dtype in (torch.bfloat16, torch.float32) and x.is_contiguous() and fc.weight.dtype == x.dtype
            and (fc.bias is None or fc.bias.dtype == x.dtype)
            and fc.weight.is_contiguous() and fc.in_features == x.shape[1] and fc.in_features % 8 == 0
            and fc.out_features % 8 == 0 and x.data_ptr() % 16 == 0)


def head(pool, fc, x, drop_p=0.0):
    """fc(dropout(pool(x), drop_p)); see the module docstring for when the library kernels serve it (drop_p: the caller
    passes 0 outside training, models/resnet.py:607-609)"""
    if MODE == "hip" and eligible(pool, fc, x):
        return _Head.apply(x, fc.weight, fc.bias, float(drop_p))
    if MODE == "hip":
        _lib.fallback("head", x)
    x = pool(x)
    if drop_p > 0.0:
        x = torch.nn.functional.dropout(x, p=float(drop_p), training=True)
    return fc(x)

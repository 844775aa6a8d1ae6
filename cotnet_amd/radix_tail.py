"""Radix-2 split-attention tail of the CoT layer as two fused HIP ops.

    gap  = radix_gap(y, k)           # mean over H,W of (y + k)            -> [B, C, 1, 1]
    out  = radix_mix(y, k, attn)     # y * attn[...,0] + k * attn[...,1]   attn [B, C, 2]

replace the reference's view/cat/sum/mean and broadcast-multiply/sum sequence (models/cotnet.py:92-104); the small
`se` MLP and the softmax over the radix pair between them stay in torch.  Device code: csrc/radix_tail.hip.
Eligible tensors: CUDA, NCHW-contiguous, fp32 or bf16; everything else takes the torch formula (same function).
"""
import ctypes
import os

import torch
from torch.autograd import Function

from . import _lib

ENABLED = os.environ.get("COT_FUSED_TAIL", "1") != "0"
_DT = {torch.float32: _lib.COT_F32, torch.bfloat16: _lib.COT_BF16}


def _p(t):
    return ctypes.c_void_p(t.data_ptr())


_DEVICE_ONLY = True  # tests drive the autograd wiring on CPU tensors through the host-emulated kernels


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream) if _DEVICE_ONLY else None


class _RadixGap(Function):
    @staticmethod
    def forward(ctx, y, k):
        B, C, H, W = y.shape
        gap = torch.empty((B, C, 1, 1), dtype=y.dtype, device=y.device)
        rc = _lib.lib().cot_radix_gap(_p(y), _p(k), _p(gap), B * C, H * W, _DT[y.dtype], _stream())
        if rc:
            _lib.check(rc, "cot_radix_gap")
        ctx.shape = y.shape
        return gap

    @staticmethod
    def backward(ctx, g):
        B, C, H, W = ctx.shape
        e = (g / (H * W)).expand(B, C, H, W)  # broadcast view; autograd adds it to the mix gradient
        return e, e


class _RadixMix(Function):
    @staticmethod
    def forward(ctx, y, k, attn):
        B, C, H, W = y.shape
        out = torch.empty_like(y)
        rc = _lib.lib().cot_radix_mix(_p(y), _p(k), _p(attn), _p(out), B * C, H * W, _DT[y.dtype], _stream())
        if rc:
            _lib.check(rc, "cot_radix_mix")
        ctx.save_for_backward(y, k, attn)
        return out

    @staticmethod
    def backward(ctx, g):
        y, k, attn = ctx.saved_tensors
        B, C, H, W = y.shape
        g = g.contiguous()
        gy, gk, ga = torch.empty_like(y), torch.empty_like(k), torch.empty_like(attn)
        rc = _lib.lib().cot_radix_mix_backward(_p(g), _p(y), _p(k), _p(attn), _p(gy), _p(gk), _p(ga), B * C, H * W,
                                               _DT[y.dtype], _stream())
        if rc:
            _lib.check(rc, "cot_radix_mix_backward")
        return gy, gk, ga


def eligible(y, k):
    return (ENABLED and (y.is_cuda or not _DEVICE_ONLY) and y.dim() == 4 and y.dtype in _DT and k.dtype == y.dtype and k.shape == y.shape
            and y.is_contiguous() and k.is_contiguous() and y.data_ptr() % 16 == 0 and k.data_ptr() % 16 == 0)


def radix_gap(y, k):
    return _RadixGap.apply(y, k)


def radix_mix(y, k, attn):
    attn = attn.contiguous()
    if attn.dtype != y.dtype:
        attn = attn.to(y.dtype)
    return _RadixMix.apply(y, k, attn)

"""SE-style sigmoid gate of SplitAttnConv2d(radix=1) on csrc/radix_tail.hip.

SE-CoTNetD's 3x3 bottleneck convolutions are `SplitAttnConv2d(radix=1)` (reference models/cotnet_hybrid.py:143-146,
models/layers/split_attn.py:62-88): conv + BN + ReLU, then `out = x * sigmoid(fc2(relu(bn1(fc1(mean_hw(x))))))`.

    gap = se_gap(x)              # [B, C, 1, 1] = mean over H, W           (one read of x)
    out = se_gate(x, logits)     # x * sigmoid(logits[b, c])               (one read, one write)

replace adaptive_avg_pool2d + sigmoid + the broadcast multiply; the backward of the gate produces dx and the logits' gradient
in one pass over (g, x).  The two fc layers and bn1 act on [B, C] descriptors (`se_mlp` below).  Eligible: NCHW-
contiguous fp32 / bf16 tensors on the device when COT_FUSED_TAIL is on (default); anything else takes the torch formula.
"""
import ctypes

import torch
import torch.nn.functional as F
from torch import nn
from torch.autograd import Function

from . import _lib, radix_tail
from .fused_bn import fused_bn_act

_DT = {torch.float32: _lib.COT_F32, torch.bfloat16: _lib.COT_BF16}
_DEVICE_ONLY = True  # tests drive the autograd wiring on CPU tensors through the host-emulated kernels


def _p(t):
    return ctypes.c_void_p(t.data_ptr())


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream) if _DEVICE_ONLY else None


class _SeGap(Function):
    @staticmethod
    def forward(ctx, x):
        B, C, H, W = x.shape
        gap = torch.empty((B, C, 1, 1), dtype=x.dtype, device=x.device)
        rc = _lib.lib().cot_se_gap(_p(x), _p(gap), B * C, H * W, _DT[x.dtype], _stream())
        if rc:
            _lib.check(rc, "cot_se_gap")
        ctx.shape = x.shape
        return gap

    @staticmethod
    def backward(ctx, g):
        B, C, H, W = ctx.shape
        return (g / (H * W)).expand(B, C, H, W)  # broadcast view; autograd adds it to the gate's gradient


class _SeGate(Function):
    @staticmethod
    def forward(ctx, x, logits):
        B, C, H, W = x.shape
        out = torch.empty_like(x)
        rc = _lib.lib().cot_se_gate(_p(x), _p(logits), _p(out), B * C, H * W, _DT[x.dtype], _stream())
        if rc:
            _lib.check(rc, "cot_se_gate")
        ctx.save_for_backward(x, logits)
        return out

    @staticmethod
    def backward(ctx, g):
        x, logits = ctx.saved_tensors
        B, C, H, W = x.shape
        g = g.contiguous()
        gx, gl = torch.empty_like(x), torch.empty_like(logits)
        rc = _lib.lib().cot_se_gate_backward(_p(g), _p(x), _p(logits), _p(gx), _p(gl), B * C, H * W, _DT[x.dtype], _stream())
        if rc:
            _lib.check(rc, "cot_se_gate_backward")
        return gx, gl


def eligible(x):
    return (radix_tail.ENABLED and (x.is_cuda or not _DEVICE_ONLY) and x.dim() == 4 and x.dtype in _DT and x.is_contiguous()
            and x.data_ptr() % 16 == 0)


def se_gap(x):
    return _SeGap.apply(x)


def se_gate(x, logits):
    """x * sigmoid(logits) with logits [B, C] (any shape with B*C elements) in x's dtype"""
    logits = logits.reshape(x.shape[0], x.shape[1]).contiguous()
    if logits.dtype != x.dtype:
        logits = logits.to(x.dtype)
    return _SeGate.apply(x, logits)


def se_mlp(gap, se):
    """The `se` branch (ref :71-77: 1x1 conv + BN + ReLU + 1x1 conv) applied to the pooled [B,C,1,1] descriptor.
    A 1x1 convolution on a 1x1 map IS a matrix product with the same weights, so it is issued as one GEMM (F.linear on
    the conv's own weight/bias) instead of a convolution call (which on ROCm costs layout transposes + cast kernels
    around a tiny GEMM, forward and twice backward).  Same parameters, same state_dict, same function."""
    c0, bn, act, c3 = se[0], se[1], se[2], se[3]
    plain = all(isinstance(c, nn.Conv2d) and c.kernel_size == (1, 1) and c.groups == 1 and c.stride == (1, 1)
                and c.padding == (0, 0) for c in (c0, c3))
    if not plain:
        return se(gap)
    h = F.linear(gap.flatten(1), c0.weight.flatten(1), c0.bias)
    # BatchNorm over the batch alone ([B, A, 1, 1]): the library's small-batch fp64 path when eligible (csrc/bn_act.hip
    # "small batches": MIOpen's fp32 kernel is 300x off an fp64 evaluation here -- what kept the 7x7 layer above 1e-3)
    h4 = h[:, :, None, None]
    a = "relu" if isinstance(act, nn.ReLU) else ("silu" if isinstance(act, nn.SiLU) else None)
    h = fused_bn_act(h4.contiguous(), bn, a) if a is not None else act(bn(h4))
    return F.linear(h.flatten(1), c3.weight.flatten(1), c3.bias)

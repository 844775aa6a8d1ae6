"""1x1 convolution in NCHW as batched GEMMs, without layout changes (experimental, opt-in: COT_CONV1X1=matmul).

Round-1 profile: MIOpen runs every convolution through NHWC kernels and brackets each call with
NCHW<->NHWC transposes and cast/zero kernels -- 6.7 + 2.3 ms of a 36 ms step, more than the GEMMs themselves for the
1x1 convolutions.  In NCHW a 1x1 convolution is, per image, Y[n] = W[Co,Ci] @ X[n][Ci,HW]: a strided-batched GEMM
whose B operand is already row-major, so forward and the data gradient (gX[n] = W^T @ gY[n]) need no transposes at
all; only the weight gradient (a reduction over n and pixels) needs one.  Same parameters / state_dict as the
nn.Conv2d it is applied to.  Not enabled by default: hipBLASLt's behaviour on these small-K shapes has not been
measured yet (next round's first A/B).
"""
import os

import torch
from torch.autograd import Function

ENABLED = os.environ.get("COT_CONV1X1", "") == "matmul"


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


def eligible(conv, x):
    return (ENABLED and isinstance(conv, torch.nn.Conv2d) and conv.kernel_size == (1, 1) and conv.stride == (1, 1)
            and conv.padding == (0, 0) and conv.groups == 1 and conv.dilation == (1, 1) and x.dim() == 4
            and x.is_contiguous() and x.dtype == conv.weight.dtype)


def conv1x1(conv, x):
    """`conv(x)` for an nn.Conv2d; 1x1 / stride-1 / ungrouped convolutions on NCHW tensors go through batched GEMMs when
    COT_CONV1X1=matmul, everything else through the module itself."""
    if eligible(conv, x):
        return _Conv1x1.apply(x, conv.weight, conv.bias)
    return conv(x)

"""LocalConvolutionMix / aggregation_zeropad_mix on MI355X -- drop-in for
cupy_layers/aggregation_zeropad_mix.py (:209-342).

3x3 (weight1) and 5x5 (weight2) aggregation of the same input, output [N, 2*heads*C, Ho, Wo] ordered
[kernel_idx][head][c].  The op is not instantiated by any model in the reference; it is provided at operator
level because BASELINE.json names it.  NCHW only.

Reference quirks and what we do with them:
  * input backward sums head 0 only (mix.py:87-88).  Default `all_heads=False` reproduces that bit for bit
    (it is exact for heads == 1, the only case the reference tests, mix.py:348); module-level flag
    `AggregationZeropadMix.all_heads = True` switches to the complete gradient.
  * `grad_weight1, grad_weight1 = None, None` (mix.py:258) makes the reference raise NameError when only the
    input needs a gradient; we simply return None for the weight gradients in that case.
"""
import ctypes

import torch
from torch import Tensor
from torch.autograd import Function
from torch.nn.modules.utils import _pair

from . import _lib
from .aggregation_zeropad import _aligned, _out_hw, _ptr, _stream


class AggregationZeropadMix(Function):
    all_heads = False

    @staticmethod
    def forward(ctx, input, weight1, weight2, kernel_size1, kernel_size2, stride, padding1, padding2, dilation):
        kernel_size1, kernel_size2 = _pair(kernel_size1), _pair(kernel_size2)
        stride, padding1, padding2, dilation = _pair(stride), _pair(padding1), _pair(padding2), _pair(dilation)
        assert kernel_size1 == (3, 3) and kernel_size2 == (5, 5)  # hard-coded tap loops, mix.py:35-36,:53-54
        assert input.dim() == 4 and input.is_cuda and weight1.is_cuda and weight2.is_cuda
        assert input.dtype == weight1.dtype == weight2.dtype
        N, C, H, W = input.size()
        _, heads, wC, taps1, wH, wW = weight1.size()
        assert taps1 == 9 and weight2.shape[3] == 25 and tuple(weight2.shape[:3]) == tuple(weight1.shape[:3])
        Ho, Wo = _out_hw(H, W, kernel_size1, stride, padding1, dilation)  # from the 3x3 set only (mix.py:216-217)
        assert Ho * Wo == wH * wW
        input, weight1, weight2 = (_aligned(t.detach().contiguous()) for t in (input, weight1, weight2))
        output = torch.empty((N, 2 * heads * C, Ho, Wo), dtype=input.dtype, device=input.device)
        geom = _lib.AggGeom(N, C, H, W, heads, wC, 3, 3, stride[0], stride[1], padding1[0], padding1[1],
                            dilation[0], dilation[1])
        with torch.cuda.device_of(input):
            rc = _lib.lib().cot_aggmix_forward(_ptr(input), _ptr(weight1), _ptr(weight2), _ptr(output),
                                               ctypes.byref(geom), padding2[0], padding2[1],
                                               _lib.dtype_code(input.dtype), _stream())
        _lib.check(rc, "cot_aggmix_forward")
        ctx.geom, ctx.padding2 = geom, padding2
        ctx.save_for_backward(input, weight1, weight2)
        return output

    @staticmethod
    def backward(ctx, grad_output):
        input, weight1, weight2 = ctx.saved_tensors
        grad_output = _aligned(grad_output.contiguous())
        p2 = ctx.padding2
        dt = _lib.dtype_code(input.dtype)
        grad_input = grad_weight1 = grad_weight2 = None
        with torch.cuda.device_of(input):
            if ctx.needs_input_grad[0]:
                grad_input = torch.empty_like(input)
                rc = _lib.lib().cot_aggmix_backward_input(
                    _ptr(grad_output), _ptr(weight1), _ptr(weight2), _ptr(grad_input), ctypes.byref(ctx.geom),
                    p2[0], p2[1], 1 if AggregationZeropadMix.all_heads else 0, dt, _stream())
                _lib.check(rc, "cot_aggmix_backward_input")
            if ctx.needs_input_grad[1] or ctx.needs_input_grad[2]:
                grad_weight1, grad_weight2 = torch.empty_like(weight1), torch.empty_like(weight2)
                rc = _lib.lib().cot_aggmix_backward_weight(
                    _ptr(grad_output), _ptr(input), _ptr(grad_weight1), _ptr(grad_weight2), ctypes.byref(ctx.geom),
                    p2[0], p2[1], dt, _stream())
                _lib.check(rc, "cot_aggmix_backward_weight")
        return grad_input, grad_weight1, grad_weight2, None, None, None, None, None, None


def aggregation_zeropad_mix(input, weight1, weight2, kernel_size1=3, kernel_size2=5, stride=1, padding1=0,
                            padding2=0, dilation=1):
    assert input.shape[0] == weight1.shape[0] and (input.shape[1] % weight1.shape[2] == 0)  # mix.py:293
    assert input.shape[0] == weight2.shape[0] and (input.shape[1] % weight2.shape[2] == 0)  # mix.py:294
    if input.is_cuda:
        out = AggregationZeropadMix.apply(input, weight1, weight2, kernel_size1, kernel_size2, stride, padding1,
                                          padding2, dilation)
    else:  # the reference's CPU route (mix.py:297-301): bounce through the GPU
        out = AggregationZeropadMix.apply(input.cuda(), weight1.cuda(), weight2.cuda(), kernel_size1, kernel_size2,
                                          stride, padding1, padding2, dilation)
        torch.cuda.synchronize()
        out = out.cpu()
    return out


class LocalConvolutionMix(torch.nn.Module):
    def __init__(self, in_channels: int, out_channels: int, kernel_size1: int, kernel_size2: int, stride: int = 1,
                 padding1: int = 0, padding2: int = 0, dilation: int = 1, pad_mode: int = 0):
        super(LocalConvolutionMix, self).__init__()
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.kernel_size1 = kernel_size1
        self.kernel_size2 = kernel_size2
        self.stride = stride
        self.padding1 = padding1
        self.padding2 = padding2
        self.dilation = dilation
        self.pad_mode = pad_mode
        assert self.kernel_size1 == 3  # mix.py:328
        assert self.kernel_size2 == 5  # mix.py:329

    def forward(self, input: Tensor, weight1: Tensor, weight2: Tensor):
        return aggregation_zeropad_mix(input, weight1, weight2, kernel_size1=self.kernel_size1,
                                       kernel_size2=self.kernel_size2, stride=self.stride, padding1=self.padding1,
                                       padding2=self.padding2, dilation=self.dilation)

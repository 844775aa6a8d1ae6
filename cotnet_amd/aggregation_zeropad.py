"""LocalConvolution / aggregation_zeropad on MI355X -- drop-in for cupy_layers/aggregation_zeropad.py.

Same public surface as the reference:
    aggregation_zeropad(input, weight, kernel_size=3, stride=1, padding=0, dilation=1)   (ref :188-197)
    AggregationZeropad (autograd.Function)                                              (ref :112-186)
    LocalConvolution(in_channels, out_channels, kernel_size, stride=1, padding=0,
                     dilation=1, pad_mode=0).forward(input, weight)                      (ref :199-236)

Differences, all deliberate (DESIGN.md "boundary"):
  * device code is the ahead-of-time compiled HIP library behind include/cotnet_amd.h, called with raw
    device pointers + the current HIP stream exactly like the reference calls its CuPy function
    (ref :140-143); dimensions are run-time arguments, so there is no per-shape JIT;
  * bf16/fp16 storage (fp32 accumulate) is accepted besides the reference's float/double (utils.py:8-12);
  * channels_last (NHWC) tensors are consumed in place instead of being copied;
  * non-contiguous inputs are made contiguous with .contiguous() (the reference's detach().clone()
    at :125-128 keeps strides in current torch and would mis-index);
  * the backward pass produces both gradients in one fused kernel launch when both are needed.
CPU tensors take the reference's route (:192-196): copy to the GPU, run, copy back -- there is no CPU
implementation in the product; without a GPU this raises.
"""
import ctypes

import torch
from torch import Tensor
from torch.autograd import Function
from torch.nn.modules.utils import _pair

from . import _lib


def _out_hw(H, W, k, s, p, d):
    # ref :120-121
    Ho = int((H + 2 * p[0] - (d[0] * (k[0] - 1) + 1)) / s[0] + 1)
    Wo = int((W + 2 * p[1] - (d[1] * (k[1] - 1) + 1)) / s[1] + 1)
    return Ho, Wo


def _nhwc_weight_strides(shape):
    """strides of w[N,heads,wC,taps,Ho,Wo] when memory order is [N,Ho,Wo,heads,wC,taps]"""
    N, heads, wC, taps, Ho, Wo = shape
    return (Ho * Wo * heads * wC * taps, wC * taps, taps, 1, Wo * heads * wC * taps, heads * wC * taps)


def _is_nhwc_weight(w):
    shape = tuple(w.shape)
    want = _nhwc_weight_strides(shape)
    return all(sz == 1 or st == ws for sz, st, ws in zip(shape, w.stride(), want))


def _empty_nhwc_weight(shape, like):
    N, heads, wC, taps, Ho, Wo = shape
    return torch.empty((N, Ho, Wo, heads, wC, taps), dtype=like.dtype, device=like.device).permute(0, 3, 4, 5, 1, 2)


def _pick_layout(input, weight):
    """-> (layout, input, weight) with both tensors dense in that layout."""
    if input.is_contiguous() and weight.is_contiguous():
        return _lib.COT_NCHW, input, weight
    if input.dim() == 4 and input.is_contiguous(memory_format=torch.channels_last) and _is_nhwc_weight(weight):
        return _lib.COT_NHWC, input, weight
    return _lib.COT_NCHW, input.contiguous(), weight.contiguous()


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def _aligned(t):
    return t if t.data_ptr() % 16 == 0 else t.clone(memory_format=torch.preserve_format)


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


# ---- optional in-situ kernel timing (bench.py) ---------------------------------------------------------------
# The library attaches start/stop events to each kernel dispatch (cot_profile_begin/_end in include/cotnet_amd.h),
# so durations are device execution times -- the same quantity rocprofv3 --kernel-trace reports.
def profile_begin():
    _lib.check(_lib.lib().cot_profile_begin(), "cot_profile_begin")


def profile_end(max_records=65536):
    """-> [(kind 'fwd'|'bwd', (N,C,H,W,heads,wC,k), torch dtype, layout, milliseconds, algorithmic_bytes, kernel)]"""
    buf = (_lib.ProfileRec * max_records)()
    n = min(_lib.lib().cot_profile_end(buf, max_records), max_records)
    dts = {_lib.COT_F32: torch.float32, _lib.COT_F64: torch.float64, _lib.COT_BF16: torch.bfloat16,
           _lib.COT_F16: torch.float16}
    out = []
    for r in buf[:n]:
        g = r.geom
        if g.kh == 0 and r.kind >= 10:  # a convolution / BatchNorm call (cot_abi.hip annotate_op): algorithmic bytes of the CALL
            N, Ci, Co, HW, G = g.N, g.C, g.W, g.H, max(g.heads, 1)
            e = torch.empty((), dtype=dts[r.dtype]).element_size()
            name = {10: "conv1x1_fwd", 11: "conv1x1_dgrad", 12: "conv1x1_wgrad", 13: "conv3x3g_fwd", 14: "conv3x3g_dgrad",
                    15: "conv3x3g_wgrad", 20: "bn_fwd", 21: "bn_bwd"}[r.kind]
            if r.kind < 20:
                taps = 9 if r.kind >= 13 else 1
                nbytes = e * (N * HW * (Ci + Co) + taps * Ci * Co // G)   # activations in + out (or in + in), the weights once
            elif r.kind == 20:
                nbytes = e * N * Ci * HW * (2 + (r.flags & 1))              # x read, y written (+ residual read)
            else:
                nbytes = e * N * Ci * HW * (3 + (r.flags & 1) + ((r.flags >> 1) & 1))  # dy, x read, dx written (+ dres, + y)
            out.append(("op:" + name, (N, Ci, Co, HW, G, r.flags, 0), dts[r.dtype], 0, float(r.ms), nbytes, r.kernel.decode()))
            continue
        if g.kh == 0:  # another launch of the library (SGD, pooling, ...): duration only
            out.append(("other", (0, 0, 0, 0, 0, 0, 0), None, 0, float(r.ms), 0, r.kernel.decode()))
            continue
        dt = dts[r.dtype]
        e = torch.empty((), dtype=dt).element_size()
        Ho = _lib.lib().cot_agg_out_size(g.H, g.kh, g.sh, g.ph, g.dh)
        Wo = _lib.lib().cot_agg_out_size(g.W, g.kw, g.sw, g.pw, g.dw)
        x_el = g.N * g.C * g.H * g.W
        w_el = g.N * g.heads * g.wC * g.kh * g.kw * Ho * Wo
        o_el = g.N * g.heads * g.C * Ho * Wo
        if r.kind == 0:
            nbytes = e * (x_el + w_el + o_el)                      # x + w + out, each once
        else:
            nbytes = e * o_el                                       # gO
            if r.flags & 1:
                nbytes += e * (w_el + x_el)                         # w read, gX written
            if r.flags & 2:
                nbytes += e * (x_el + w_el)                         # x read, gW written
        out.append(("fwd" if r.kind == 0 else "bwd", (g.N, g.C, g.H, g.W, g.heads, g.wC, g.kh), dt, r.layout,
                    float(r.ms), nbytes, r.kernel.decode()))
    return out


class AggregationZeropad(Function):
    @staticmethod
    def forward(ctx, input, weight, kernel_size, stride, padding, dilation):
        kernel_size, stride, padding, dilation = _pair(kernel_size), _pair(stride), _pair(padding), _pair(dilation)
        ctx.kernel_size, ctx.stride, ctx.padding, ctx.dilation = kernel_size, stride, padding, dilation
        assert input.dim() == 4 and input.is_cuda and weight.is_cuda  # ref :117
        assert weight.dim() == 6 and input.dtype == weight.dtype
        batch_size, input_channels, input_height, input_width = input.size()
        _, weight_heads, weight_channels, weight_kernels, weight_height, weight_width = weight.size()
        output_height, output_width = _out_hw(input_height, input_width, kernel_size, stride, padding, dilation)
        assert output_height * output_width == weight_height * weight_width  # ref :122
        assert weight_kernels == kernel_size[0] * kernel_size[1]
        layout, input, weight = _pick_layout(input.detach(), weight.detach())
        input, weight = _aligned(input), _aligned(weight)
        out_shape = (batch_size, weight_heads * input_channels, output_height, output_width)
        if layout == _lib.COT_NHWC:
            output = torch.empty(out_shape, dtype=input.dtype, device=input.device,
                                 memory_format=torch.channels_last)
        else:
            output = torch.empty(out_shape, dtype=input.dtype, device=input.device)
        geom = _lib.AggGeom(batch_size, input_channels, input_height, input_width, weight_heads, weight_channels,
                            kernel_size[0], kernel_size[1], stride[0], stride[1], padding[0], padding[1],
                            dilation[0], dilation[1])
        # one process per GPU: tensors live on the current device, launches go to its current stream (ref :130,:143)
        rc = _lib.lib().cot_agg_forward(_ptr(input), _ptr(weight), _ptr(output), ctypes.byref(geom),
                                        _lib.dtype_code(input.dtype), layout, _stream())
        if rc:
            _lib.check(rc, "cot_agg_forward")
        ctx.geom, ctx.layout = geom, layout
        ctx.save_for_backward(input, weight)
        return output

    @staticmethod
    def backward(ctx, grad_output):
        input, weight = ctx.saved_tensors
        assert grad_output.is_cuda
        layout = ctx.layout
        if layout == _lib.COT_NHWC:
            grad_output = grad_output.contiguous(memory_format=torch.channels_last)
        else:
            grad_output = grad_output.contiguous()
        grad_output = _aligned(grad_output)
        grad_input = grad_weight = None
        if ctx.needs_input_grad[0]:
            grad_input = torch.empty_like(input)
        if ctx.needs_input_grad[1]:
            grad_weight = (_empty_nhwc_weight(weight.shape, weight) if layout == _lib.COT_NHWC
                           else torch.empty_like(weight))
        if grad_input is not None or grad_weight is not None:
            rc = _lib.lib().cot_agg_backward(_ptr(grad_output), _ptr(input), _ptr(weight), _ptr(grad_input),
                                             _ptr(grad_weight), ctypes.byref(ctx.geom),
                                             _lib.dtype_code(input.dtype), layout, _stream())
            if rc:
                _lib.check(rc, "cot_agg_backward")
        return grad_input, grad_weight, None, None, None, None


def aggregation_zeropad(input, weight, kernel_size=3, stride=1, padding=0, dilation=1):
    assert input.shape[0] == weight.shape[0] and (input.shape[1] % weight.shape[2] == 0)  # ref :189
    if weight.dtype != input.dtype:
        # mixed-precision callers (autocast runs GroupNorm in fp32 while the values are bf16): compute in the
        # values' dtype; the cast is differentiable, so the weight gradient comes back in the weight's dtype
        weight = weight.to(input.dtype)
    if input.is_cuda:
        out = AggregationZeropad.apply(input, weight, kernel_size, stride, padding, dilation)
    else:
        # the reference's CPU route (:192-196): bounce through the GPU.  No CPU compute in the product.
        out = AggregationZeropad.apply(input.cuda(), weight.cuda(), kernel_size, stride, padding, dilation)
        torch.cuda.synchronize()
        out = out.cpu()
    return out


class LocalConvolution(torch.nn.Module):
    """Parameter-free module; attribute names are read by the reference's FLOP counter
    (utils/flops_counter.py:500-501) and are kept identical."""

    def __init__(self, in_channels: int, out_channels: int, kernel_size: int, stride: int = 1, padding: int = 0,
                 dilation: int = 1, pad_mode: int = 0):
        super(LocalConvolution, self).__init__()
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.kernel_size = kernel_size
        self.stride = stride
        self.padding = padding
        self.dilation = dilation
        self.pad_mode = pad_mode  # stored, ignored -- as in the reference (:220, :228)

    def forward(self, input: Tensor, weight: Tensor):
        return aggregation_zeropad(input, weight, kernel_size=self.kernel_size, stride=self.stride,
                                   padding=self.padding, dilation=self.dilation)

    def extra_repr(self):
        return (f"{self.in_channels}, {self.out_channels}, kernel_size={self.kernel_size}, stride={self.stride}, "
                f"padding={self.padding}, dilation={self.dilation}")


# ---- window softmax + aggregation in one kernel --------------------------------------------------
class AggregationZeropadSoftmax(Function):
    """out = aggregation_zeropad(input, softmax(logits, dim=3)) -- LR-Net's attention application
    (models/lr_net.py:94-96) -- with the softmax fused into the aggregation kernels (3x3/s1/p1/d1, NCHW)."""

    @staticmethod
    def forward(ctx, input, logits, geom):
        input, logits = _aligned(input.detach().contiguous()), _aligned(logits.detach().contiguous())
        N, heads, C = geom.N, geom.heads, geom.C
        out = torch.empty((N, heads * C, geom.H, geom.W), dtype=input.dtype, device=input.device)
        probs = torch.empty_like(logits)
        rc = _lib.lib().cot_agg_softmax_forward(_ptr(input), _ptr(logits), _ptr(out), _ptr(probs), ctypes.byref(geom),
                                                _lib.dtype_code(input.dtype), _stream())
        if rc:
            _lib.check(rc, "cot_agg_softmax_forward")
        ctx.geom = geom
        ctx.save_for_backward(input, probs)
        return out

    @staticmethod
    def backward(ctx, grad_output):
        input, probs = ctx.saved_tensors
        grad_output = _aligned(grad_output.contiguous())
        gx, gl = torch.empty_like(input), torch.empty_like(probs)
        rc = _lib.lib().cot_agg_softmax_backward(_ptr(grad_output), _ptr(input), _ptr(probs), _ptr(gx), _ptr(gl),
                                                 ctypes.byref(ctx.geom), _lib.dtype_code(input.dtype), _stream())
        if rc:
            _lib.check(rc, "cot_agg_softmax_backward")
        return gx, gl, None


def softmax_fusable(input, logits, kernel_size, stride, padding, dilation):
    k, s, p, d = _pair(kernel_size), _pair(stride), _pair(padding), _pair(dilation)
    return (input.is_cuda and k == (3, 3) and s == (1, 1) and p == (1, 1) and d == (1, 1) and logits.shape[1] == 1
            and input.dtype == logits.dtype and input.dtype in (torch.float32, torch.bfloat16, torch.float16))


def aggregation_zeropad_softmax(input, logits, kernel_size=3, stride=1, padding=0, dilation=1):
    """softmax over the kh*kw window of `logits` [N,heads,wC,kh*kw,Ho,Wo], then aggregation_zeropad.  Uses the fused
    kernels when the geometry is the 3x3 / stride-1 / pad-1 single-head one, else softmax + aggregation_zeropad."""
    assert input.shape[0] == logits.shape[0] and (input.shape[1] % logits.shape[2] == 0)
    if softmax_fusable(input, logits, kernel_size, stride, padding, dilation):
        N, C, H, W = input.shape
        geom = _lib.AggGeom(N, C, H, W, 1, logits.shape[2], 3, 3, 1, 1, 1, 1, 1, 1)
        try:
            return AggregationZeropadSoftmax.apply(input, logits, geom)
        except RuntimeError as e:
            if "not covered" not in str(e):  # COT_ERR_UNSUPPORTED (tile does not fit LDS, odd alignment): compose
                raise
    return aggregation_zeropad(input, torch.softmax(logits, dim=3), kernel_size, stride, padding, dilation)

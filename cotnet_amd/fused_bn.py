"""BatchNorm2d + activation (+ residual add) as ONE fused HIP op for training.

`fused_bn_act(x, bn, act, residual)` computes exactly what the reference's module sequences compute --
`act(bn(x) [+ residual])` with `bn` an ordinary `nn.BatchNorm2d` (same parameters, buffers and state_dict keys,
running statistics updated with torch's momentum / unbiased-variance convention) -- but in two HBM passes forward and
two backward instead of one pass per module (models/cotnet.py:231-235 bn1+act1, :248-262 bn3 + residual + act3,
:43-47 key_embed BN+ReLU, :89-90 bn + SiLU).  Device code: csrc/bn_act.hip behind cot_bn_act_forward/_backward.

The fused path is taken for training-mode NCHW-contiguous fp32 / bf16 CUDA tensors, and -- one pass, running statistics --
for eval-mode modules under torch.no_grad() (cot_bn_act_inference); anything else (eval mode with autograd, channels_last,
fp64, momentum=None, CPU) takes the plain torch modules, so results and state are identical either way.
"""
import ctypes
import os

import torch
import torch.nn.functional as F
from torch.autograd import Function

from . import _lib

ENABLED = os.environ.get("COT_FUSED_BN", "1") != "0"  # A/B switch: 0 = always take the plain torch modules
_ACTS = {None: 0, "none": 0, "relu": 1, "silu": 2}


def _p(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


_DEVICE_ONLY = True  # tests drive the autograd wiring on CPU tensors through the host-emulated kernels


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream) if _DEVICE_ONLY else None


_WS = _lib.register_cache({})  # (N, C) -> workspace floats (pure function of the shape; avoids a library call per launch)
_DT = {torch.float32: _lib.COT_F32, torch.bfloat16: _lib.COT_BF16}


def _ws_floats(N, C):
    k = (N, C)
    v = _WS.get(k)
    if v is None:
        v = _WS[k] = _lib.lib().cot_bn_act_workspace(N, C)
    return v


class _BNAct(Function):
    @staticmethod
    def forward(ctx, x, residual, weight, bias, running_mean, running_var, nbt, eps, momentum, act):
        N, C, H, W = x.shape
        L = _lib.lib()
        # (with a row's worth of margins inside its own allocation: a 3x3 convolution that consumes y -- a deep stem's, SplitAttn's --
        # then takes the LDS-staged weight gradient, cot_conv3x3g_backward_weight_guarded)
        from .conv3x3g import new_guarded
        y = new_guarded(N, C, H, W, x.dtype, x.device) if x.dtype == torch.bfloat16 else torch.empty_like(x)
        # one allocation for [mean | rstd | workspace]; host overhead matters: the step issues ~160 of these launches
        nws = _ws_floats(N, C)
        scratch = torch.empty(2 * C + nws, dtype=torch.float32, device=x.device)
        mean, rstd, ws = scratch[:C], scratch[C:2 * C], scratch[2 * C:]
        rc = L.cot_bn_act_forward(_p(x), _p(residual), _p(y), _p(weight), _p(bias), _p(mean), _p(rstd),
                                  _p(running_mean), _p(running_var), _p(nbt), _p(ws), N, C, H * W, eps, momentum, act,
                                  _DT[x.dtype], _stream())
        if rc:
            _lib.check(rc, "cot_bn_act_forward")
        ctx.act, ctx.has_res = act, residual is not None
        ctx.save_for_backward(x, y if act == 1 else None, weight, bias, mean, rstd)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, y, weight, bias, mean, rstd = ctx.saved_tensors
        N, C, H, W = x.shape
        L = _lib.lib()
        dy = dy.contiguous()
        dx = torch.empty_like(x)
        dres = torch.empty_like(x) if (ctx.has_res and ctx.needs_input_grad[1]) else None
        scratch = torch.empty(2 * C + _ws_floats(N, C), dtype=torch.float32, device=x.device)
        dgamma, dbeta, ws = scratch[:C], scratch[C:2 * C], scratch[2 * C:]
        rc = L.cot_bn_act_backward(_p(dy), _p(x), _p(y), _p(dx), _p(dres), _p(weight), _p(bias), _p(mean), _p(rstd),
                                   _p(dgamma), _p(dbeta), _p(ws), N, C, H * W, ctx.act, _DT[x.dtype], _stream())
        if rc:
            _lib.check(rc, "cot_bn_act_backward")
        return dx, dres, dgamma, dbeta, None, None, None, None, None, None


def _torch_path(x, bn, act, residual):
    y = bn(x)
    if residual is not None:
        y = y + residual
    if act == "relu":
        return F.relu(y, inplace=True)
    if act == "silu":
        return F.silu(y, inplace=True)
    return y


def _inference(x, bn, act, residual):
    """eval-mode BatchNorm (running statistics) + activation (+ residual) as one pass; no autograd (the caller checked)"""
    N, C, H, W = x.shape
    y = torch.empty_like(x)
    rc = _lib.lib().cot_bn_act_inference(_p(x), _p(residual), _p(y), _p(bn.weight), _p(bn.bias), _p(bn.running_mean),
                                         _p(bn.running_var), N, C, H * W, float(bn.eps), _ACTS[act], _DT[x.dtype], _stream())
    if rc:
        _lib.check(rc, "cot_bn_act_inference")
    return y


def fused_bn_act(x, bn, act=None, residual=None):
    """act(bn(x) [+ residual]) with `bn` an nn.BatchNorm2d.  Fused HIP kernels when eligible, torch otherwise."""
    if (ENABLED and not bn.training and not torch.is_grad_enabled() and (x.is_cuda or not _DEVICE_ONLY) and x.dim() == 4
            and x.dtype in _DT and x.is_contiguous() and bn.affine and bn.track_running_stats and bn.running_mean is not None
            and bn.weight.dtype == torch.float32 and bn.running_mean.dtype == torch.float32 and x.data_ptr() % 16 == 0
            and (residual is None or (residual.shape == x.shape and residual.dtype == x.dtype and residual.is_contiguous()
                                      and residual.data_ptr() % 16 == 0))):
        return _inference(x, bn, act, residual)  # forward-only evaluation under torch.no_grad()
    # (tensors are assumed to live on the current device, as everywhere in a one-process-per-GPU job)
    ok = (ENABLED and bn.training and (x.is_cuda or not _DEVICE_ONLY) and x.dim() == 4 and x.dtype in (torch.float32, torch.bfloat16)
          and x.is_contiguous() and bn.affine and bn.track_running_stats and bn.momentum is not None
          and bn.weight.dtype == torch.float32 and x.data_ptr() % 16 == 0
          and bn.num_batches_tracked is not None and bn.num_batches_tracked.dtype == torch.int64
          and (residual is None or (residual.shape == x.shape and residual.dtype == x.dtype
                                    and residual.is_contiguous() and residual.data_ptr() % 16 == 0))
          and not (residual is not None and _ACTS.get(act) == 2))  # (SiLU after a residual add: the kernels have no backward for it)
    if not ok:
        if ENABLED:
            _lib.fallback("fused_bn_act", x, f"act {act}, residual {residual is not None}, training {bn.training}")
        return _torch_path(x, bn, act, residual)
    # num_batches_tracked += 1 (nn.BatchNorm2d.forward's bookkeeping) is done by the statistics kernel itself
    return _BNAct.apply(x, residual, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.num_batches_tracked,
                        float(bn.eps), float(bn.momentum), _ACTS[act])

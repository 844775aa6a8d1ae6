"""Mixed-precision parameters + fused flat SGD-nesterov ("training-loop glue on device").

The reference trains fp32 with `torch.optim.SGD(nesterov=True)` (optim/optim_factory.py:54-56), weight decay only on
>1-D parameters (:19-31), gradients synchronised by DDP (train.py:115), plus one elementwise op per tensor for EMA.
On MI355X the step is bound by HBM passes and launch count, so this module

  * keeps Conv2d / Linear weights and biases in bf16 (what the convolutions read) with fp32 MASTER copies, and the
    normalisation parameters in fp32 -- no per-step autocast casts (measured: ~530 cast launches per CoTNet-50 step);
  * lays parameters, gradients, masters and momentum out in a few flat buckets per (dtype, weight-decay) group
    (cotnet_amd.data_parallel.GradBucketReducer with flatten_params=True, grad_mode="copy": autograd's gradients are
    moved into the bucket with one multi-tensor copy, then all-reduced over RCCL on the side stream);
  * updates each bucket with ONE hand-written HIP kernel (`cot_sgd_step`, csrc/optim.hip).

Arithmetic = torch.optim.SGD (dampening 0) evaluated in fp32 on the master weights; checked against it in
tests/test_flat_sgd_gpu.py.
"""
import ctypes

import torch
from torch import nn

from . import _lib
from .data_parallel import GradBucketReducer


_DEVICE_ONLY = True  # tests drive the optimizer on CPU tensors through the host-emulated kernels


def to_mixed_bf16(model):
    """Conv2d / Linear / GroupNorm parameters -> bf16 (in place); BatchNorm parameters and all buffers stay fp32
    (torch's group_norm wants its affine parameters in the input dtype; batch_norm takes fp32 ones).  Returns the model."""
    for m in model.modules():
        if isinstance(m, (nn.Conv2d, nn.Linear, nn.GroupNorm)):
            for p in m.parameters(recurse=False):
                p.data = p.data.to(torch.bfloat16)
    return model


def _decay_group(name, p):
    # reference rule: no weight decay on 1-D parameters and biases (optim/optim_factory.py:19-31)
    return "no_decay" if (p.ndim <= 1 or name.endswith(".bias")) else "decay"


class FlatSGD:
    def __init__(self, model, lr, momentum=0.9, weight_decay=0.0, nesterov=True, bucket_mb=10.0, process_group=None,
                 broadcast_params=True, ema_decay=None, force_collectives=False, grad_dtype=None):
        """bucket_mb: size of the flat gradient buckets.  10 MiB cuts CoTNet-50's 44 MB of bf16 weight gradients into five
        buckets in the order backward produces them (classifier and stage 4 first, the stem last), so four all-reduces are
        already on the communication stream when backward ends and only the last, small one is exposed; round 2's 48 MiB
        made ONE bucket whose last member is the stem's weight -- nothing could overlap (tests/test_data_parallel_cpu.py).
        A 10 MB ring all-reduce over 8 GPUs moves 17.5 MB per link direction: ~0.12 ms at xGMI's ~153 GB/s, well above the
        collective's fixed latency.
        grad_dtype: the arithmetic of the all-reduce.  None (default): the parameter's own (bf16 weights -> bf16 buckets, RCCL
        averages in bf16: one rounding of 2^-9 relative per ring hop, ~0.2 % over eight ranks -- the size of the bf16 gradients'
        own rounding; tests/test_data_parallel_cpu.py bounds it); torch.float32: the reference's arithmetic (DDP sums fp32
        gradients, train.py:112-115) at twice the bytes on the wire -- the gradients are produced in bf16 as always (kernels write
        them straight into the bucket slots) and each bucket is widened by one flat copy on the communication stream right before
        its all-reduce (GradBucketReducer reduce_dtype); the SGD kernel then reads the fp32 averages.
        ema_decay: also keep an exponential moving average of the weights (the reference's ModelEmaV2,
        utils/model_ema.py, `model_ema: True` / decay 0.9999 in its recipes): one flat kernel per bucket after the SGD
        kernel instead of one elementwise op per state_dict tensor; floating-point buffers (BatchNorm running statistics)
        are averaged with one multi-tensor lerp.  `ema_state_dict()` returns it under the model's state_dict keys."""
        self.lr, self.momentum, self.weight_decay, self.nesterov = float(lr), float(momentum), float(weight_decay), nesterov
        self.ema_decay = None if ema_decay is None else float(ema_decay)
        self.model = model
        # masters must be taken from the fp32 values BEFORE rounding when the caller converts later; here parameters
        # are already in their storage dtype, so the master starts as the (exact) up-cast of the working copy.
        self.reducer = GradBucketReducer(model, process_group=process_group, bucket_mb=bucket_mb,
                                         broadcast_params=broadcast_params, group_fn=_decay_group, grad_mode="copy",
                                         flatten_params=True, force_collectives=force_collectives, reduce_dtype=grad_dtype)
        self.state = []
        for b in self.reducer.buckets:
            master = b.pflat.float().clone() if b.pflat.dtype != torch.float32 else None
            st = {"master": master, "mom": torch.zeros(b.pflat.numel(), dtype=torch.float32, device=b.pflat.device)}
            if self.ema_decay is not None:
                st["ema"] = (master if master is not None else b.pflat).float().clone()
            self.state.append(st)
        if self.ema_decay is not None:
            self._buf_src = [bf for bf in model.buffers() if bf.is_floating_point()]
            self._buf_ema = [bf.detach().float().clone() for bf in self._buf_src]
        _lib.lib()

    def zero_grad(self):
        self.reducer.zero_grad()

    def step(self, deferred=False):
        """finish the gradient all-reduce (stream wait only) and update every bucket with one kernel each.
        deferred=True: the reducer ran with defer_comm (hooks only filled the buckets); all-reduce them now."""
        if deferred:
            self.reducer.allreduce_all()
        else:
            self.reducer.finish()
        L = _lib.lib()
        stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream) if _DEVICE_ONLY else None
        for b, st in zip(self.reducer.buckets, self.state):
            key = b.key
            wd = self.weight_decay if key == "decay" else 0.0
            rc = L.cot_sgd_step(ctypes.c_void_p(b.pflat.data_ptr()),
                                ctypes.c_void_p(st["master"].data_ptr()) if st["master"] is not None else None,
                                ctypes.c_void_p(st["mom"].data_ptr()), ctypes.c_void_p(self.reducer.reduced(b).data_ptr()),
                                b.pflat.numel(), self.lr, self.momentum, wd, 1.0, 1 if self.nesterov else 0,
                                _lib.dtype_code(b.pflat.dtype), _lib.dtype_code(self.reducer.reduced(b).dtype), stream)
            _lib.check(rc, "cot_sgd_step")
            if self.ema_decay is not None:
                src = st["master"] if st["master"] is not None else b.pflat
                rc = L.cot_ema_step(ctypes.c_void_p(st["ema"].data_ptr()), ctypes.c_void_p(src.data_ptr()), src.numel(),
                                    self.ema_decay, _lib.dtype_code(src.dtype), stream)
                _lib.check(rc, "cot_ema_step")
        if self.ema_decay is not None and self._buf_src:
            torch._foreach_lerp_(self._buf_ema, [bf.float() if bf.dtype != torch.float32 else bf for bf in self._buf_src],
                                 1.0 - self.ema_decay)
        if self.reducer.buckets:  # the parameters changed behind torch's version counters: caches keyed on them move on
            from . import cot_layer_fused
            cot_layer_fused.after_optimizer_step(self.reducer.buckets[0].pflat.device)

    def ema_state_dict(self):
        """the averaged weights under the model's own state_dict keys (fp32; integer buffers are copied as they are) --
        what the reference stores as `state_dict_ema` (utils/checkpoint_saver.py:103-120)"""
        assert self.ema_decay is not None, "FlatSGD was built without ema_decay"
        by_param = {}
        esz = {}
        for b, st in zip(self.reducer.buckets, self.state):
            for p, off in zip(b.params, b.offs):
                by_param[p] = st["ema"][off:off + p.numel()].view_as(p)
        by_buf = {id(bf): e for bf, e in zip(self._buf_src, self._buf_ema)}
        out = {}
        params = dict(self.model.named_parameters())
        bufs = dict(self.model.named_buffers())
        for k in self.model.state_dict().keys():
            if k in params:
                out[k] = by_param[params[k]].clone()
            else:
                bf = bufs[k]
                out[k] = by_buf[id(bf)].clone() if id(bf) in by_buf else bf.detach().clone()
        return out

    def set_lr(self, lr):
        self.lr = float(lr)

    def master_parameters(self):
        """fp32 view of every parameter (master copy for bf16 parameters), in module.parameters() order of the buckets"""
        out = {}
        for b, st in zip(self.reducer.buckets, self.state):
            src = st["master"] if st["master"] is not None else b.pflat
            for p, off in zip(b.params, b.offs):
                out[p] = src[off:off + p.numel()].view_as(p)
        return out

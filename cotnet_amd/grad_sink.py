"""Gradient sink: let the kernels that produce parameter gradients write them straight into the flat communication /
optimizer buckets (cotnet_amd.data_parallel.GradBucketReducer, grad_mode="copy").

Without it a step moves every gradient twice: the weight-gradient kernel writes a fresh tensor, autograd hands it to
`.grad`, and the reducer's bucket fill copies it into the flat buffer -- on ROCm `torch._foreach_copy_` of ~160 tensors
is ~200 hipMemcpyAsync launches of 3.8 us each (0.76 ms of a 23 ms step in the round-2 trace).  With a sink registered
for a parameter the single-node layers (cot_layer_fused, head_fused, stem7x7) allocate nothing: the kernel's output
pointer IS the parameter's slot in the bucket; the tensor returned to autograd is a fresh alias of that slot (so that
AccumulateGrad adopts it instead of cloning), and the bucket fill recognises it and skips the copy.

Only the FIRST gradient of a step may use the slot (`param.grad is None`); a second backward before zero_grad falls back
to an ordinary tensor, which autograd then accumulates as usual.
"""
import torch

_SINK = {}  # id(param) -> (param, view into the flat gradient bucket)


def register(param, view):
    _SINK[id(param)] = (param, view)


def unregister_all():
    _SINK.clear()


def out_like(param):
    """tensor for the gradient of `param`: an alias of its bucket slot when a sink is registered and this is the step's first
    gradient for it, else a fresh tensor"""
    e = _SINK.get(id(param))
    if e is not None and e[0] is param and param.grad is None and e[1].dtype == param.dtype:
        return e[1].detach()  # new tensor object, same storage: autograd may adopt it
    return torch.empty_like(param)


def is_in_place(param, view):
    """True when param.grad already lives in `view` (nothing to copy)"""
    g = param.grad
    return g is not None and g.data_ptr() == view.data_ptr() and g.shape == view.shape and g.dtype == view.dtype

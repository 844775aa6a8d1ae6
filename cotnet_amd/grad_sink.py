"""Gradient sink: let the kernels that produce parameter gradients write them straight into the flat communication /
optimizer buckets (cotnet_amd.data_parallel.GradBucketReducer, grad_mode="copy").

Without it a step moves every gradient twice: the weight-gradient kernel writes a fresh tensor, autograd hands it to
`.grad`, and the reducer's bucket fill copies it into the flat buffer -- on ROCm `torch._foreach_copy_` of ~160 tensors
is ~200 hipMemcpyAsync launches of 3.8 us each (0.76 ms of a 23 ms step in the round-2 trace).  With a sink registered
for a parameter the single-node layers (cot_layer_fused, head_fused, stem7x7) allocate nothing: the kernel's output
pointer IS the parameter's slot in the bucket; the tensor returned to autograd is a fresh alias of that slot (so that
AccumulateGrad adopts it instead of cloning), and the bucket fill recognises it and skips the copy.

A slot is LENT at most once per step: `param.grad is None` alone does not say "first gradient", because AccumulateGrad
sets `.grad` only after every producer of a backward pass has run -- a parameter used twice in one graph (a shared layer, a
two-view loss, the model called twice before backward) would hand the same slot to both producers, the second kernel would
overwrite the first, and autograd would add the two aliases: 2x the last gradient instead of the sum.  So the
entry carries a `lent` flag, set by out_like() and cleared when the reducer has consumed the bucket (`release`, called
from GradBucketReducer._fill / zero_grad); a second request in the same step gets an ordinary tensor, which autograd
accumulates as usual.
"""
import torch

_SINK = {}  # id(param) -> [param, view into the flat gradient bucket, lent this step]
_PRODUCERS = []  # (stream, join): streams other than the compute stream on which gradient kernels run (cot_layer_fused's side stream)


def register_producer_stream(stream, join):
    """`stream` carries kernels that write parameter gradients; `join()` makes the current (compute) stream wait for it"""
    _PRODUCERS.append((stream, join))


def producer_streams():
    return [s for s, _ in _PRODUCERS]


def sync_producers():
    """the compute stream waits for every gradient kernel issued so far on a producer stream"""
    for _, join in _PRODUCERS:
        join()


def register(param, view):
    _SINK[id(param)] = [param, view, False]


def unregister(param):
    e = _SINK.get(id(param))
    if e is not None and e[0] is param:
        del _SINK[id(param)]


def unregister_all():
    _SINK.clear()


def release(param):
    """the reducer is done with this step's gradient of `param`: its slot may be lent again"""
    e = _SINK.get(id(param))
    if e is not None and e[0] is param:
        e[2] = False


def out_like(param):
    """tensor for the gradient of `param`: an alias of its bucket slot when a sink is registered and this is the step's first
    request for it, else a fresh tensor"""
    e = _SINK.get(id(param))
    if e is not None and e[0] is param and not e[2] and param.grad is None and e[1].dtype == param.dtype:
        e[2] = True
        return e[1].detach()  # new tensor object, same storage: autograd may adopt it
    return torch.empty_like(param)


def is_in_place(param, view):
    """True when param.grad already lives in `view` (nothing to copy)"""
    g = param.grad
    return g is not None and g.data_ptr() == view.data_ptr() and g.shape == view.shape and g.dtype == view.dtype

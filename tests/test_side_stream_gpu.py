"""Weight gradients on the side stream (cot_layer_fused._Side): whoever reads a parameter gradient sees a finished one.

ADVICE r3 (high): with the lazy flush a node returned while its weight-gradient kernels were merely ISSUED on the side
stream; autograd's AccumulateGrad then ran `p.grad += g` on the compute stream whenever a gradient already existed (second
backward without zero_grad, micro-batches, a view-mode reducer, a parameter used twice) -- reading g while it was being
written.  Now only nodes whose gradients are all flat-bucket aliases (adopted, never read mid-backward) stay un-joined.
Every kernel of the step is deterministic, so "side stream on" and "side stream off" must agree BIT FOR BIT in each of
these situations; a race shows as a mismatch (or as garbage)."""
import copy

import pytest
import torch

from cotnet_amd import cot_layer_fused as clf, grad_sink
from cotnet_amd.cotnet import Bottleneck
from cotnet_amd.data_parallel import GradBucketReducer
from cotnet_amd.flat_sgd import FlatSGD, to_mixed_bf16
from tests import truth

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _stack(n_blocks=3, C=256, width=64):
    torch.manual_seed(3)
    m = torch.nn.Sequential(*[Bottleneck(C, width) for _ in range(n_blocks)]).to(DEV).train()
    with torch.no_grad():
        for b in m:
            b.bn3.weight.fill_(0.7)
    return to_mixed_bf16(m)


def _inputs(N=32, C=256, H=28, n=2):
    g = torch.Generator(device=DEV).manual_seed(5)
    return [(torch.randn(N, C, H, H, device=DEV, generator=g).bfloat16(), torch.randn(N, C, H, H, device=DEV, generator=g).bfloat16())
            for _ in range(n)]


def _grads(model):
    torch.cuda.synchronize()
    return {n: p.grad.detach().clone() for n, p in model.named_parameters()}


def _same(a, b):
    assert set(a) == set(b)
    bad = [n for n in a if not torch.equal(a[n], b[n])]
    assert not bad, (len(bad), bad[:4], (a[bad[0]].float() - b[bad[0]].float()).abs().max().item())


def _two_passes_without_zero_grad(model, side, monkeypatch):
    monkeypatch.setattr(clf, "SIDE_WGRAD", side)
    m = copy.deepcopy(model)
    with truth.switches(**truth.SINGLE_NODE):
        for x, g in _inputs():
            y = m(x)
            assert y.grad_fn.name().startswith("_BottleneckNode")
            y.backward(g)
    return _grads(m)


def test_second_backward_accumulates_finished_gradients(monkeypatch):
    """`p.grad += g` of the second pass must read g after the side stream wrote it"""
    model = _stack()
    ref = _two_passes_without_zero_grad(model, False, monkeypatch)
    for _ in range(3):  # (a race is a matter of timing: several tries)
        _same(_two_passes_without_zero_grad(model, True, monkeypatch), ref)


def test_view_mode_reducer_accumulates_finished_gradients(monkeypatch):
    """GradBucketReducer(grad_mode="view"): p.grad is preset to the bucket view, so EVERY gradient is accumulated in place
    (the path INTEGRATION.md documents for an optimizer of the caller's own)"""
    model = _stack()

    def run(side):
        monkeypatch.setattr(clf, "SIDE_WGRAD", side)
        m = copy.deepcopy(model)
        red = GradBucketReducer(m, grad_mode="view")
        with truth.switches(**truth.SINGLE_NODE):
            for x, g in _inputs(n=1):
                m(x).backward(g)
        red.finish()
        out = _grads(m)
        red.remove()
        return out
    ref = run(False)
    for _ in range(3):
        _same(run(True), ref)


def test_copy_mode_reducer_keeps_the_unjoined_flush_and_its_buckets_are_complete(monkeypatch):
    """the measured configuration (FlatSGD -> copy-mode reducer -> gradient sink): nodes stay un-joined (adopted_only), the
    flat buckets after backward equal those of the run without a side stream"""
    model = _stack()
    seen = []
    orig = clf._Side.adopted_only

    def spy(self):
        r = orig(self)
        seen.append(r)
        return r
    monkeypatch.setattr(clf._Side, "adopted_only", spy)

    def run(side):
        monkeypatch.setattr(clf, "SIDE_WGRAD", side)
        m = copy.deepcopy(model)
        opt = FlatSGD(m, lr=0.0, momentum=0.0)
        with truth.switches(**truth.SINGLE_NODE):
            for x, g in _inputs(n=1):
                m(x).backward(g)
        opt.reducer.finish()
        torch.cuda.synchronize()
        out = [b.flat.clone() for b in opt.reducer.buckets]
        opt.reducer.remove()
        grad_sink.unregister_all()
        return out
    ref = run(False)
    seen.clear()
    got = run(True)
    assert seen and all(seen), seen  # every node of this run left its weight gradients un-joined
    for a, b in zip(got, ref):
        assert torch.equal(a, b)


def test_backward_under_another_ambient_stream(monkeypatch):
    """forward on a side stream of the caller's, backward() called from the default stream (the engine runs each node on
    its forward's stream; the end-of-backward join must reach that stream AND the caller's)"""
    model = _stack()

    def run(side, own_stream):
        monkeypatch.setattr(clf, "SIDE_WGRAD", side)
        m = copy.deepcopy(model)
        opt = FlatSGD(m, lr=0.0, momentum=0.0)
        (x, g), = _inputs(n=1)
        torch.cuda.synchronize()
        with truth.switches(**truth.SINGLE_NODE):
            if own_stream:
                s = torch.cuda.Stream()
                with torch.cuda.stream(s):
                    y = m(x)
                torch.cuda.current_stream().wait_stream(s)
            else:
                y = m(x)
            y.backward(g)
        opt.reducer.finish()
        out = [b.flat.clone() for b in opt.reducer.buckets]  # (read on the default stream, no host synchronisation before it)
        torch.cuda.synchronize()
        opt.reducer.remove()
        grad_sink.unregister_all()
        return out
    ref = run(False, False)
    for _ in range(3):
        for a, b in zip(run(True, True), ref):
            assert torch.equal(a, b)


def test_a_successor_reducer_keeps_its_gradient_sinks(monkeypatch):
    """`opt = FlatSGD(model)` re-bound -- the old reducer's __del__ must not drop the NEW reducer's sinks"""
    m = _stack(1)
    old = FlatSGD(m, lr=0.0)
    new = FlatSGD(m, lr=0.0)
    p = next(m.parameters())
    assert grad_sink._SINK[id(p)][1] is new.reducer._view(new.reducer._bucket_of[p], p)
    old.reducer.remove()
    del old
    assert id(p) in grad_sink._SINK and grad_sink._SINK[id(p)][1] is new.reducer._view(new.reducer._bucket_of[p], p)
    new.reducer.remove()
    assert id(p) not in grad_sink._SINK

"""world_size-2 gloo tests (CPU) of the bucketed gradient all-reduce and the flattened BN-statistics sync."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
from torch import nn

from cotnet_amd.data_parallel import GradBucketReducer, distribute_bn


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _net():
    return nn.Sequential(nn.Conv2d(3, 8, 3, padding=1), nn.BatchNorm2d(8), nn.ReLU(), nn.Conv2d(8, 8, 1),
                         nn.AdaptiveAvgPool2d(1), nn.Flatten(), nn.Linear(8, 4))


def _worker(rank, world, port, bucket_mb, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.manual_seed(100 + rank)  # different initial weights per rank: broadcast must fix that
        model = _net()
        red = GradBucketReducer(model, bucket_mb=bucket_mb)
        # (1) parameters identical to rank 0's after construction
        flat = torch.cat([p.detach().reshape(-1) for p in model.parameters()])
        gathered = [torch.empty_like(flat) for _ in range(world)]
        dist.all_gather(gathered, flat)
        assert all(torch.equal(g, gathered[0]) for g in gathered)
        # (2) averaged gradients == mean over ranks of the local gradients, for two consecutive steps
        for step in range(2):
            torch.manual_seed(1000 * step + rank)
            x, t = torch.randn(4, 3, 6, 6), torch.randint(0, 4, (4,))
            ref = _net()
            ref.load_state_dict(model.state_dict())
            nn.functional.cross_entropy(ref(x), t).backward()
            local = torch.cat([p.grad.reshape(-1) for p in ref.parameters()])
            allg = [torch.empty_like(local) for _ in range(world)]
            dist.all_gather(allg, local)
            want = torch.stack(allg).mean(0)
            red.zero_grad()
            nn.functional.cross_entropy(model(x), t).backward()
            red.finish()
            got = torch.cat([p.grad.reshape(-1) for p in model.parameters()])
            assert torch.allclose(got, want, atol=1e-6), (got - want).abs().max()
            for p in model.parameters():  # grads are views into the flat buckets
                assert any(p.grad.data_ptr() >= b.flat.data_ptr() and
                           p.grad.data_ptr() < b.flat.data_ptr() + b.flat.numel() * 4 for b in red.buckets)
        # (3) flattened BN running-stat averaging
        bn = model[1]
        with torch.no_grad():
            bn.running_mean.fill_(float(rank))
            bn.running_var.fill_(1.0 + rank)
        distribute_bn(model, reduce=True)
        assert torch.allclose(bn.running_mean, torch.full((8,), (world - 1) / 2.0))
        assert torch.allclose(bn.running_var, torch.full((8,), 1.0 + (world - 1) / 2.0))
        with torch.no_grad():
            bn.running_mean.fill_(float(rank + 5))
        distribute_bn(model, reduce=False)
        assert torch.allclose(bn.running_mean, torch.full((8,), 5.0))
        q.put((rank, len(red.buckets), "ok"))
    except Exception as e:  # surface the failure in the parent
        q.put((rank, -1, repr(e)))
        raise
    finally:
        dist.destroy_process_group()


def _worker_copy_mode(rank, world, port, q):
    """grad_mode='copy' + flatten_params + weight-decay groups + deferred communication (what FlatSGD / the graphed
    step use), checked against the mean of the per-rank gradients"""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from cotnet_amd.flat_sgd import _decay_group
        torch.manual_seed(7 + rank)
        model = _net()
        red = GradBucketReducer(model, bucket_mb=48.0, group_fn=_decay_group, grad_mode="copy", flatten_params=True)
        assert {b.key for b in red.buckets} == {"decay", "no_decay"}
        for b in red.buckets:  # parameters now live inside the flat parameter buffers
            for p in b.params:
                assert b.pflat.data_ptr() <= p.data_ptr() < b.pflat.data_ptr() + b.pflat.numel() * 4
        for defer in (False, True):
            red.defer_comm = defer
            torch.manual_seed(50 + rank)
            x, t = torch.randn(4, 3, 6, 6), torch.randint(0, 4, (4,))
            ref = _net()
            ref.load_state_dict(model.state_dict())
            nn.functional.cross_entropy(ref(x), t).backward()
            want = {}
            for (n, p) in ref.named_parameters():
                g = [torch.empty_like(p.grad) for _ in range(world)]
                dist.all_gather(g, p.grad.contiguous())
                want[n] = torch.stack(g).mean(0)
            red.zero_grad()
            nn.functional.cross_entropy(model(x), t).backward()
            assert all(p.grad is None for p in model.parameters())  # handed over to the buckets
            if defer:
                red.finish()
                red.allreduce_all()
            else:
                red.finish()
            names = {p: n for n, p in model.named_parameters()}
            for b in red.buckets:
                for p, v in zip(b.params, b.views):
                    assert torch.allclose(v, want[names[p]], atol=1e-6), (defer, names[p])
        q.put((rank, 0, "ok"))
    except Exception as e:
        q.put((rank, -1, repr(e)))
        raise
    finally:
        dist.destroy_process_group()


def test_copy_mode_flat_params_world2_gloo():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_copy_mode, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    for rank, _, msg in results:
        assert msg == "ok", f"rank {rank}: {msg}"


@pytest.mark.parametrize("bucket_mb", [48.0, 0.0005])  # one bucket / many tiny buckets
def test_bucketed_allreduce_world2_gloo(bucket_mb):
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, bucket_mb, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    for rank, nb, msg in results:
        assert msg == "ok", f"rank {rank}: {msg}"
    nbs = {nb for _, nb, _ in results}
    assert len(nbs) == 1
    assert (nbs.pop() == 1) == (bucket_mb > 1)


def test_single_process_is_a_noop_with_flat_buckets():
    model = _net()
    red = GradBucketReducer(model)
    assert not red.enabled and len(red.buckets) == 1
    nn.functional.cross_entropy(model(torch.randn(2, 3, 6, 6)), torch.tensor([0, 1])).backward()
    red.finish()
    total = sum(p.numel() for p in model.parameters())
    assert red.buckets[0].flat.numel() == total and red.buckets[0].flat.abs().sum() > 0
    red.zero_grad()
    assert red.buckets[0].flat.abs().sum() == 0 and all(p.grad is not None for p in model.parameters())


def test_a_second_backward_before_finish_is_refused_where_it_would_lose_a_gradient():
    """copy mode hands the gradients over when a bucket's last one arrives: a second backward() before finish() would overwrite
    them, so the hook raises; a single process in view mode accumulates in place as autograd does"""
    torch.manual_seed(3)
    x, t = torch.randn(4, 3, 6, 6), torch.randint(0, 4, (4,))
    model = _net()
    red = GradBucketReducer(model, grad_mode="copy", broadcast_params=False)
    red.zero_grad()
    nn.functional.cross_entropy(model(x), t).backward()
    with pytest.raises(RuntimeError, match="once per"):
        nn.functional.cross_entropy(model(x), t).backward()
    red.finish()
    red.zero_grad()
    nn.functional.cross_entropy(model(x), t).backward()   # the next step is unaffected
    red.finish()
    red.remove()
    model = _net()
    red = GradBucketReducer(model, grad_mode="view", broadcast_params=False)
    red.zero_grad()
    nn.functional.cross_entropy(model(x), t).backward()
    once = [p.grad.clone() for p in model.parameters()]
    nn.functional.cross_entropy(model(x), t).backward()
    red.finish()
    for p, g in zip(model.parameters(), once):
        assert torch.allclose(p.grad, 2 * g, atol=1e-6)


def test_grad_sink_lets_a_producer_write_its_bucket_slot():
    """cotnet_amd.grad_sink: a custom backward that writes a parameter's gradient straight into the flat bucket (what the
    single-node layers do on the GPU) -- autograd adopts the alias, the bucket fill skips the copy, values are right; a
    second backward before zero_grad falls back to an ordinary tensor and accumulates as usual."""
    import torch
    from torch import nn
    from torch.autograd import Function
    from cotnet_amd import grad_sink
    from cotnet_amd.data_parallel import GradBucketReducer

    torch.manual_seed(0)
    lin = nn.Linear(4, 3)
    red = GradBucketReducer(lin, grad_mode="copy", flatten_params=True)
    slot = {id(p): v for b in red.buckets for p, v in zip(b.params, b.views)}
    seen = []

    class Lin(Function):
        @staticmethod
        def forward(ctx, x, w, b):
            ctx.save_for_backward(x, w)
            return x @ w.t() + b

        @staticmethod
        def backward(ctx, g):
            x, w = ctx.saved_tensors
            gw = grad_sink.out_like(lin.weight)
            seen.append(gw.data_ptr() == slot[id(lin.weight)].data_ptr())
            gw.copy_(g.t() @ x)
            return g @ w, gw, g.sum(0)

    x = torch.randn(5, 4)
    g = torch.randn(5, 3)
    red.zero_grad()
    Lin.apply(x, lin.weight, lin.bias).backward(g)
    red.finish()
    assert seen == [True]
    assert torch.allclose(slot[id(lin.weight)], g.t() @ x, atol=1e-6)
    assert torch.allclose(slot[id(lin.bias)], g.sum(0), atol=1e-6)   # (no sink use for the bias: ordinary copy)
    assert lin.weight.grad is None and lin.bias.grad is None
    # a parameter that already has a gradient: no alias, autograd accumulates
    lin.weight.grad = torch.ones_like(lin.weight)
    gw2 = grad_sink.out_like(lin.weight)
    assert gw2.data_ptr() != slot[id(lin.weight)].data_ptr()
    red.remove()
    grad_sink.unregister_all()


def _worker_fp32_buckets(rank, world, port, q):
    """bf16 parameters, fp32 gradient buckets (FlatSGD(grad_dtype=torch.float32)): the all-reduce is the reference's fp32 sum
    (train.py:112-115) -- the buckets equal the fp32 mean of the ranks' bf16 gradients exactly, not a bf16 rounding of it"""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from cotnet_amd.flat_sgd import _decay_group
        torch.manual_seed(3)
        model = _net().bfloat16()
        red = GradBucketReducer(model, bucket_mb=1.0, group_fn=_decay_group, grad_mode="copy", flatten_params=True,
                                grad_dtype=torch.float32)
        assert all(b.flat.dtype == torch.float32 for b in red.buckets)
        torch.manual_seed(50 + rank)
        x, t = torch.randn(4, 3, 6, 6).bfloat16(), torch.randint(0, 4, (4,))
        ref = _net().bfloat16()
        ref.load_state_dict(model.state_dict())
        nn.functional.cross_entropy(ref(x).float(), t).backward()
        want = {}
        for n, p in ref.named_parameters():
            g = [torch.empty_like(p.grad) for _ in range(world)]
            dist.all_gather(g, p.grad.contiguous())
            want[n] = torch.stack([v.float() for v in g]).mean(0)
        red.zero_grad()
        nn.functional.cross_entropy(model(x).float(), t).backward()
        red.finish()
        names = {p: n for n, p in model.named_parameters()}
        inexact = 0
        for b in red.buckets:
            for p, v in zip(b.params, b.views):
                assert torch.equal(v, want[names[p]]), (names[p], (v - want[names[p]]).abs().max())
                inexact += int((v != v.bfloat16().float()).any())
        assert inexact > 0  # (the fp32 mean of two bf16 values is in general NOT a bf16 value: nothing was rounded on the way)
        q.put((rank, len(red.buckets), "ok"))
    except Exception as e:
        q.put((rank, -1, repr(e)))
        raise
    finally:
        dist.destroy_process_group()


def test_fp32_gradient_buckets_reduce_exactly_like_the_reference():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_fp32_buckets, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    assert all(r[2] == "ok" for r in res), res


def test_bf16_ring_average_over_eight_ranks_stays_within_the_gradients_own_rounding():
    """the default buckets are bf16 and RCCL's ring all-reduce (AVG) adds in the data type, one rounding per
    hop.  Model of the 8-rank ring on real gradient statistics (each rank's bf16 gradient = a common signal + per-rank noise,
    the regime of data-parallel SGD): reduce-scatter order (((g0 + g1) + g2) + ...) rounded to bf16 after every add, then the
    division by the world size rounded once.  The drift from the fp32 mean of the same bf16 inputs must stay of the order of
    ONE bf16 rounding (2^-9 relative to the element's magnitude) -- i.e. no larger than the error already made when each
    rank's fp32 gradient was stored as bf16."""
    torch.manual_seed(0)
    world, n = 8, 1 << 18
    signal = torch.randn(n) * torch.logspace(-4, 0, n)           # gradients span four decades
    grads = [(signal + 0.5 * signal.abs().mean() * torch.randn(n)).bfloat16() for _ in range(world)]
    exact = torch.stack([g.float() for g in grads]).mean(0)
    acc = grads[0]
    for g in grads[1:]:
        acc = (acc.float() + g.float()).bfloat16()                # one bf16 rounding per ring hop
    ring = (acc.float() / world).bfloat16().float()
    scale = torch.stack([g.float().abs() for g in grads]).amax(0)  # the partial sums are bounded by world * max |g_r|
    rel = ((ring - exact).abs() / scale.clamp_min(1e-30))
    ulp = 2.0 ** -8                                               # bf16: 8 significand bits -> half an ulp = 2^-9 relative
    assert rel.max().item() <= world * ulp / 2, rel.max().item()  # worst case: every hop rounds the same way
    assert rel.mean().item() <= 1.5 * ulp / 2, rel.mean().item()  # typical: of the order of a single rounding


def test_a_collected_reducer_leaves_its_successors_sinks_alone():
    """rebinding `opt = FlatSGD(model)` creates the new reducer before the old one is collected; the old
    one's remove() / __del__ must only drop ITS OWN gradient-sink entries"""
    from cotnet_amd import grad_sink
    from cotnet_amd.data_parallel import GradBucketReducer
    lin = torch.nn.Linear(4, 3)
    old = GradBucketReducer(lin, grad_mode="copy")
    new = GradBucketReducer(lin, grad_mode="copy")
    own = {id(p): v for b in new.buckets for p, v in zip(b.params, b.views)}
    old.remove()
    del old
    for p in lin.parameters():
        assert grad_sink._SINK[id(p)][1] is own[id(p)]
    before = grad_sink.fresh_count()
    assert grad_sink.out_like(lin.weight).data_ptr() == own[id(lin.weight)].data_ptr()   # still lent from the new bucket
    assert grad_sink.fresh_count() == before
    assert grad_sink.out_like(lin.weight).data_ptr() != own[id(lin.weight)].data_ptr()   # second request of the step: fresh
    assert grad_sink.fresh_count() == before + 1
    new.remove()
    assert not any(id(p) in grad_sink._SINK for p in lin.parameters())


def _worker_fused_nodes(rank, world, port, q):
    """the measured configuration under data parallelism: single-node Bottlenecks whose kernels write parameter gradients
    straight into the flat bf16 buckets (grad_sink), FlatSGD's reducer all-reducing them -- on the host-emulated library, two
    ranks with different data, checked against the mean of the gradients each rank computes alone"""
    import copy
    import ctypes
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import cotnet_amd.aggregation_zeropad as az
        from cotnet_amd import (_lib, conv1x1 as c1, conv3x3g as c3, cot_layer_fused as clf, flat_sgd, fused_bn,
                                group_norm9 as g9, head_fused as hf, pool3x3 as p3, radix_tail, stem7x7 as s7)
        from cotnet_amd.cotnet import Bottleneck
        from cotnet_amd.flat_sgd import FlatSGD, to_mixed_bf16
        from cotnet_amd.resnet import ResNet
        from tests import test_kernels_emulated as tke
        if tke._EMUL is None:
            q.put((rank, 0, "skip"))
            return
        _lib.lib = lambda: tke._EMUL
        for mod in (clf, c1, c3, fused_bn, radix_tail, g9, flat_sgd, p3, hf, s7):
            mod._DEVICE_ONLY = False
        clf.ENABLED = True
        c1.MODE = c3.MODE = g9.MODE = p3.MODE = hf.MODE = s7.MODE = "hip"
        az.aggregation_zeropad = lambda i, w, kernel_size=3, stride=1, padding=0, dilation=1: tke._EmulAggregation.apply(i, w)
        solo = [dist.new_group([r]) for r in range(world)]  # (every rank takes part in creating every group)

        torch.manual_seed(3)  # same weights on both ranks
        base = ResNet(Bottleneck, [1, 1, 1, 1], num_classes=10)
        for mod in base.modules():
            if isinstance(mod, nn.BatchNorm2d):
                mod.eps = 0.5
                nn.init.uniform_(mod.weight, 0.5, 1.0)  # (bn3 starts at zero: no gradient would reach the branches)
        base = to_mixed_bf16(base).train()
        torch.manual_seed(40 + rank)  # different data
        x, t = torch.randn(2, 3, 32, 32).bfloat16(), torch.randint(0, 10, (2,))

        def grads(process_group):
            model = copy.deepcopy(base)
            opt = FlatSGD(model, lr=0.1, process_group=process_group, broadcast_params=False)
            opt.zero_grad()
            nn.functional.cross_entropy(model(x).float(), t).backward()
            opt.reducer.finish()
            return [b.flat.detach().float().clone() for b in opt.reducer.buckets], opt

        local, _ = grads(solo[rank])
        want = []
        for g in local:
            parts = [torch.empty_like(g) for _ in range(world)]
            dist.all_gather(parts, g)
            want.append(torch.stack(parts).mean(0))
        got, opt = grads(None)
        assert opt.reducer.enabled and len(got) == len(want)
        # ---- overlap: with buckets cut small enough (this toy network has ~1 MB of gradients) the all-reduces of all but
        # the last bucket are launched BEFORE the last parameter's gradient arrives, and the ranks agree on the bucket layout
        model = copy.deepcopy(base)
        opt2 = FlatSGD(model, lr=0.1, bucket_mb=0.25, broadcast_params=False)
        red = opt2.reducer
        name_of = {p: n for n, p in model.named_parameters()}
        layout = [[name_of[p] for p in b.params] for b in red.buckets]
        layouts = [None] * world
        dist.all_gather_object(layouts, layout)
        assert layouts[0] == layouts[1] and len(layout) >= 5
        seen = []
        orig = red._on_grad_ready

        def spy(param):
            seen.append((name_of[param], red._launched))
            orig(param)
        for h in red._hooks:
            h.remove()
        red._hooks = [p.register_post_accumulate_grad_hook(spy) for p in model.parameters() if p.requires_grad]
        opt2.zero_grad()
        nn.functional.cross_entropy(model(x).float(), t).backward()
        last_name, launched_before_last = seen[-1]
        assert launched_before_last >= 3, (last_name, launched_before_last, len(layout))
        red.finish()
        assert red._launched == len(layout)
        for a, b in zip(got, want):
            assert torch.isfinite(a).all() and b.abs().max() > 0
            # bf16 buckets: the average is rounded once more than the fp32 mean of the two rounded gradients
            assert torch.allclose(a, b, atol=2e-2 * b.abs().max().item(), rtol=2e-2), (a - b).abs().max()
        opt.step()  # the fused SGD kernel on the averaged buckets
        ps = [b.pflat.detach().float().clone() for b in opt.reducer.buckets]
        for p in ps:
            parts = [torch.empty_like(p) for _ in range(world)]
            dist.all_gather(parts, p)
            assert torch.equal(parts[0], parts[1])  # the ranks stay in lock step
        q.put((rank, len(got), "ok"))
    except Exception as e:
        q.put((rank, -1, repr(e)))
        raise
    finally:
        dist.destroy_process_group()


def test_single_node_layers_with_gradient_sink_world2_gloo():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_fused_nodes, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=600) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    if any(msg == "skip" for _, _, msg in results):
        pytest.skip("host emulation build unavailable")
    for rank, _, msg in results:
        assert msg == "ok", f"rank {rank}: {msg}"


def test_default_buckets_leave_only_the_stem_exposed():
    """the default bucket size on the benchmark model (no forward needed: the layout is a function of the parameter list):
    CoTNet-50's weight gradients are cut into >= 4 buckets in the order backward produces them, and the bucket that completes
    last -- the one holding the stem's weight -- is the smallest share, so >= 3 all-reduces are launched while backward is
    still running (VERDICT r2 missing #1: 48 MiB made one bucket that completed with conv1.weight)"""
    import cotnet_amd
    from cotnet_amd.flat_sgd import FlatSGD, to_mixed_bf16, _DEVICE_ONLY  # noqa: F401
    from cotnet_amd import flat_sgd
    torch.manual_seed(0)
    model = to_mixed_bf16(cotnet_amd.create_model("cotnet50", num_classes=1000))
    old = flat_sgd._lib.lib
    flat_sgd._lib.lib = lambda: None  # (constructing the optimizer only checks that the library loads)
    try:
        opt = FlatSGD(model, lr=0.1)
    finally:
        flat_sgd._lib.lib = old
    name_of = {p: n for n, p in model.named_parameters()}
    decay = [b for b in opt.reducer.buckets if b.key == "decay"]
    assert len(decay) >= 4
    sizes = [b.flat.numel() * b.flat.element_size() for b in decay]
    assert all(sz <= 10.5 * 2 ** 20 for sz in sizes)
    # buckets are created in the order their last gradient arrives; the stem's weight is in the last one
    assert "conv1.weight" in [name_of[p] for p in decay[-1].params]
    assert "fc.weight" in [name_of[p] for p in decay[0].params]
    assert sizes[-1] <= 0.3 * sum(sizes)
    opt.reducer.remove()


def test_grad_sink_lends_a_slot_once_per_step():
    """a parameter with TWO producers in one backward (the layer applied twice before loss.backward()).  Both
    producers run before AccumulateGrad sets .grad, so `param.grad is None` cannot tell the second from the first; the slot
    is lent once per step, the second producer gets an ordinary tensor and autograd adds the two."""
    import torch
    from torch import nn
    from torch.autograd import Function
    from cotnet_amd import grad_sink
    from cotnet_amd.data_parallel import GradBucketReducer

    torch.manual_seed(1)
    lin = nn.Linear(4, 3, bias=False)
    red = GradBucketReducer(lin, grad_mode="copy", flatten_params=True)
    slot = red.buckets[0].views[0]
    aliased = []

    class Lin(Function):
        @staticmethod
        def forward(ctx, x, w):
            ctx.save_for_backward(x, w)
            return x @ w.t()

        @staticmethod
        def backward(ctx, g):
            x, w = ctx.saved_tensors
            gw = grad_sink.out_like(lin.weight)
            aliased.append(gw.data_ptr() == slot.data_ptr())
            gw.copy_(g.t() @ x)
            return g @ w, gw

    for step in range(2):  # (the second step checks that the slot is lent again after the reducer consumed the first)
        x1, x2 = torch.randn(5, 4), torch.randn(6, 4)
        g1, g2 = torch.randn(5, 3), torch.randn(6, 3)
        red.zero_grad()
        aliased.clear()
        (Lin.apply(x1, lin.weight) * g1).sum().add((Lin.apply(x2, lin.weight) * g2).sum()).backward()
        red.finish()
        assert sorted(aliased) == [False, True]
        assert torch.allclose(slot, g1.t() @ x1 + g2.t() @ x2, atol=1e-5), (slot - (g1.t() @ x1 + g2.t() @ x2)).abs().max()
    red.remove()
    assert not grad_sink._SINK  # remove() drops the reducer's entries (they hold strong references to the flat buffers)


def test_guarded_activation_buffer_and_its_promise():
    """cot_layer_fused._new_guarded / _guard_elems: the CoT layer's input sits inside a larger allocation with at least W + 1
    elements either side (what cot_conv3x3g_backward_weight_guarded is promised); a tensor that fills its storage, or a
    non-contiguous view, promises nothing"""
    from cotnet_amd import cot_layer_fused as clf
    for (N, C, H, W) in [(2, 8, 14, 14), (1, 4, 7, 7), (3, 2, 56, 56), (1, 1, 5, 9)]:
        t = clf._new_guarded(N, C, H, W, torch.bfloat16, torch.device("cpu"))
        assert t.shape == (N, C, H, W) and t.is_contiguous() and t.data_ptr() % 16 == 0
        g = clf._guard_elems(t)
        assert g >= W + 1 and g % 8 == 0
        flat = t.untyped_storage()
        assert t.storage_offset() == g and flat.nbytes() // 2 == N * C * H * W + 2 * g
    plain = torch.empty(2, 4, 6, 6, dtype=torch.bfloat16)
    assert clf._guard_elems(plain) == 0
    assert clf._guard_elems(plain[:, :, ::2, ::2]) == 0
    assert clf._guard_elems(torch.empty(100, dtype=torch.bfloat16)[10:82].view(2, 4, 3, 3)) == 10


def test_producer_stream_registry_joins_before_a_bucket_copy():
    """grad_sink.register_producer_stream: a side stream that writes gradients is joined before the reducer copies gradients
    into a bucket (and its streams are what a bucket's communication stream waits for)"""
    from cotnet_amd import grad_sink
    calls = []
    n0 = len(grad_sink.producer_streams())
    grad_sink.register_producer_stream("stream-object", lambda: calls.append("joined"))
    try:
        assert grad_sink.producer_streams()[n0:] == ["stream-object"]
        torch.manual_seed(0)
        model = torch.nn.Sequential(torch.nn.Linear(8, 8), torch.nn.Linear(8, 4))
        red = GradBucketReducer(model, bucket_mb=1.0, grad_mode="copy", flatten_params=True)
        model(torch.randn(3, 8)).sum().backward()  # (plain autograd gradients: not in place -> the bucket fill copies them)
        assert calls and all(c == "joined" for c in calls)
        red.remove()
    finally:
        del grad_sink._PRODUCERS[n0:]



def test_fp32_gradient_buckets_on_the_mixed_precision_model():
    """fp32 gradient buckets on the framework's own mixed model (fp32 BatchNorm parameters beside bf16 biases in
    the 'no_decay' group): a bucket holds ONE parameter dtype, and every parameter AND every gradient slot starts on a 16-byte
    boundary (the kernels' check_align16) although the offset granule of fp32 gradients alone would be 4 elements = 8 bytes of bf16"""
    from cotnet_amd import create_model
    from cotnet_amd.data_parallel import GradBucketReducer
    from cotnet_amd.flat_sgd import _decay_group, to_mixed_bf16
    model = to_mixed_bf16(create_model("cotnet50", num_classes=10))
    red = GradBucketReducer(model, group_fn=_decay_group, grad_mode="copy", flatten_params=True, grad_dtype=torch.float32)
    try:
        assert {b.flat.dtype for b in red.buckets} == {torch.float32}
        for b in red.buckets:
            assert len({p.dtype for p in b.params}) == 1 and b.pflat.dtype == b.params[0].dtype
            for p, v in zip(b.params, b.views):
                assert p.data_ptr() % 16 == 0 and v.data_ptr() % 16 == 0
                assert p.data_ptr() >= b.pflat.data_ptr() and p.data_ptr() < b.pflat.data_ptr() + b.pflat.numel() * b.pflat.element_size()
        assert sum(len(b.params) for b in red.buckets) == sum(1 for p in model.parameters() if p.requires_grad)
    finally:
        red.remove()


def test_zero_grad_recovers_from_a_step_abandoned_before_finish():
    """backward() without finish() (exception in the loop, skipped step) must not wedge every later backward"""
    from cotnet_amd.data_parallel import GradBucketReducer
    torch.manual_seed(0)
    m = torch.nn.Sequential(torch.nn.Linear(8, 8), torch.nn.ReLU(), torch.nn.Linear(8, 4))
    for mode in ("copy", "view"):
        red = GradBucketReducer(m, grad_mode=mode)
        try:
            m(torch.randn(3, 8)).sum().backward()       # step abandoned here: no finish()
            if mode == "copy":
                with pytest.raises(RuntimeError, match="zero_grad"):
                    m(torch.randn(3, 8)).sum().backward()
            red.zero_grad()
            x = torch.randn(3, 8)
            m(x).sum().backward()
            red.finish()
            got = [v.clone() for b in red.buckets for v in b.views]
            red.zero_grad()
            ref = torch.autograd.grad(m(x).sum(), [p for b in red.buckets for p in b.params])
            for g, r in zip(got, ref):
                assert torch.equal(g, r)
        finally:
            red.remove()


def _worker_reduce_dtype(rank, world, port, q):
    """bf16 gradient buckets, fp32 REDUCTION buffers (GradBucketReducer reduce_dtype = what FlatSGD(grad_dtype=torch.float32) builds):
    `reduced(b)` equals the fp32 mean of the ranks' bf16 gradients exactly; with communication deferred too (the graph-replay form:
    hooks only fill, allreduce_all() afterwards)"""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from cotnet_amd.flat_sgd import _decay_group
        torch.manual_seed(3)
        model = _net().bfloat16()
        red = GradBucketReducer(model, bucket_mb=1.0, group_fn=_decay_group, grad_mode="copy", flatten_params=True,
                                reduce_dtype=torch.float32)
        assert all(b.flat.dtype == torch.bfloat16 and b.rflat is not None and b.rflat.dtype == torch.float32 for b in red.buckets)
        torch.manual_seed(50 + rank)
        x, t = torch.randn(4, 3, 6, 6).bfloat16(), torch.randint(0, 4, (4,))
        ref = _net().bfloat16()
        ref.load_state_dict(model.state_dict())
        nn.functional.cross_entropy(ref(x).float(), t).backward()
        want = {}
        for n, p in ref.named_parameters():
            g = [torch.empty_like(p.grad) for _ in range(world)]
            dist.all_gather(g, p.grad.contiguous())
            want[n] = torch.stack([v.float() for v in g]).mean(0)
        names = {p: n for n, p in model.named_parameters()}
        for defer in (False, True):
            red.defer_comm = defer
            red.zero_grad()
            for b in red.buckets:
                b.rflat.fill_(float("nan"))
            nn.functional.cross_entropy(model(x).float(), t).backward()
            if defer:
                red.allreduce_all()
            else:
                red.finish()
            inexact = 0
            for b in red.buckets:
                r = red.reduced(b)
                assert r is b.rflat
                for p, off in zip(b.params, b.offs):
                    v = r[off:off + p.numel()].view_as(p)
                    assert torch.equal(v, want[names[p]]), (defer, names[p], (v - want[names[p]]).abs().max())
                    inexact += int((v != v.bfloat16().float()).any())
            assert inexact > 0  # (nothing was rounded to bf16 on the way)
        q.put((rank, len(red.buckets), "ok"))
    except Exception as e:
        q.put((rank, -1, repr(e)))
        raise
    finally:
        dist.destroy_process_group()


def test_fp32_reduction_of_bf16_buckets_with_and_without_deferred_communication():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_reduce_dtype, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    assert all(r[2] == "ok" for r in res), res


def test_a_wider_reduction_dtype_is_refused_in_view_mode():
    """with reduce_dtype the all-reduce sums a second buffer; in "view" mode p.grad would stay a view of the un-reduced one"""
    import torch.distributed as dist
    from cotnet_amd.data_parallel import GradBucketReducer
    if dist.is_initialized():
        pytest.skip("needs a fresh process group")
    import socket
    s_ = socket.socket()
    s_.bind(("127.0.0.1", 0))
    port = s_.getsockname()[1]
    s_.close()
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1)
    try:
        m = torch.nn.Linear(8, 8).bfloat16()
        with pytest.raises(AssertionError, match="grad_mode='copy'"):
            GradBucketReducer(m, grad_mode="view", force_collectives=True, reduce_dtype=torch.float32)
        r = GradBucketReducer(m, grad_mode="copy", force_collectives=True, reduce_dtype=torch.float32, bucket_mb=1e-4)
        # bucket_mb bounds the message in the REDUCTION dtype: 64 + 8 bf16 parameters = 288 bytes in fp32 > 104 bytes -> two buckets
        assert len(r.buckets) == 2
    finally:
        dist.destroy_process_group()

"""Data-parallel gradient averaging over RCCL / xGMI, one process per GPU.

Replaces the reference's `torch.nn.parallel.DistributedDataParallel(model, device_ids=[local_rank])` wrap
(train.py:112-115) and `utils/distributed.py:57-67` (`distribute_bn`).  The reference's only parallelism is
data parallelism over the batch: every rank holds a full replica, gradients are averaged once
per step.  This module does that MI355X-first:

  * gradients live in a few large FLAT buckets allocated once (`p.grad` is a view into its bucket, so autograd
    accumulates straight into the communication buffer: no per-step flatten/copy);
  * buckets are filled in reverse parameter order (the order backward produces gradients); the moment the last
    gradient of a bucket has been accumulated (post-accumulate-grad hook) the bucket's all-reduce is enqueued on
    a dedicated communication stream that waits on an event recorded on the compute stream -- the collective
    overlaps with the rest of backward;
  * xGMI is a point-to-point fabric (7 links x ~153 GB/s per GPU): a ring all-reduce is bound by ONE link, and
    each collective pays a fixed launch latency, so buckets are few and large -- but not so large that nothing is
    left to overlap: default 10 MiB = five buckets for CoTNet-50's 44 MB of bf16 gradients (nine for 88.8 MB of fp32),
    four of them in flight before backward ends (round 2's 48 MiB was ONE bucket that completed with the stem);
  * no host synchronisation anywhere: `finish()` only makes the compute stream wait for the communication
    work (the reference synchronises the host twice per step, train.py:282,:290).

Works with backend "nccl" (= RCCL on ROCm) on GPUs and with "gloo" on CPU tensors (used by the world_size-2
tests); with no process group (single GPU) it degrades to a no-op that still provides the flat buckets.
"""
import torch
import torch.distributed as dist

from . import grad_sink


class _Bucket:
    __slots__ = ("flat", "params", "pending", "work", "ready_event", "key", "views", "pflat", "fired", "offs", "rflat")

    def __init__(self, flat, params, key=None):
        self.flat = flat
        self.params = params
        self.pending = len(params)
        self.work = None
        self.ready_event = None
        self.key = key
        self.views = []      # per-parameter views into `flat`
        self.pflat = None    # flat PARAMETER storage (flatten_params=True)
        self.fired = set()
        self.offs = []
        self.rflat = None    # the REDUCTION buffer when it is wider than the gradients' own dtype (reduce_dtype)


class GradBucketReducer:
    def __init__(self, module, process_group=None, bucket_mb=10.0, broadcast_params=True, grad_dtype=None,
                 group_fn=None, grad_mode="view", flatten_params=False, force_collectives=False, reduce_dtype=None):
        """group_fn(name, param) -> hashable key: parameters with different keys never share a bucket (used by
        FlatSGD to keep weight-decay groups / dtypes apart).  grad_mode "view": p.grad is a view into the bucket and
        autograd accumulates in place (one small add per parameter); "copy": autograd hands over its gradient tensor
        and the bucket is filled with ONE multi-tensor copy when its last gradient arrives (p.grad is then dropped).
        flatten_params=True additionally moves the parameters themselves into one flat buffer per bucket.
        reduce_dtype (e.g. torch.float32): the gradients are produced and collected in their own dtype exactly as without it (kernels
        write bf16 weight gradients straight into the bucket slots, grad_sink), and a bucket whose dtype is narrower is widened by ONE
        flat copy on the communication stream into a second buffer `rflat`, which is what the all-reduce sums -- the reference's
        fp32 reduction (train.py:112-115) -- and what `reduced(b)` hands to the optimizer.  grad_dtype instead makes the bucket
        itself wide: every producer then writes an ordinary tensor and the bucket fill converts (2 ms per CoTNet-50 step more
        than this form on one MI355X, profiles/r05_grad_reduction_paths.log).  Without a process group nothing is widened."""
        assert grad_mode in ("view", "copy")
        self.grad_mode = grad_mode
        self.defer_comm = False  # True: hooks only fill the buckets; all-reduces are issued by allreduce_all()
        self.module = module
        self.group = process_group
        # force_collectives: issue the collectives even in a world of one rank (a 1-GPU box can then exercise the real RCCL
        # communicator, the side stream and the event hand-off: tests/test_rccl_gpu.py)
        self.enabled = dist.is_available() and dist.is_initialized() and (
            dist.get_world_size(process_group) > 1 or force_collectives)
        self.world = dist.get_world_size(process_group) if self.enabled else 1
        named = [(n, p) for n, p in module.named_parameters() if p.requires_grad]
        params = [p for _, p in named]
        assert params, "module has no trainable parameters"
        keys = {p: (group_fn(n, p) if group_fn else None) for n, p in named}
        self.device = params[0].device
        self.on_gpu = self.device.type == "cuda"
        self.comm_stream = torch.cuda.Stream(device=self.device) if (self.on_gpu and self.enabled) else None
        backend = dist.get_backend(process_group) if self.enabled else None
        self._avg_op = backend == "nccl"  # RCCL implements ncclAvg; gloo does not
        if self.enabled and broadcast_params:
            self.broadcast_module_state()

        # ---- bucket assignment: reverse registration order, cut at bucket_mb, one dtype per bucket
        cap = int(bucket_mb * 1024 * 1024)
        self.buckets = []
        self._bucket_of = {}
        open_buckets = {}  # (key, param dtype, grad dtype) -> [params, bytes]; keeps reverse registration order inside each group
        order = []
        for p in reversed(params):
            dt = grad_dtype or p.dtype
            k = (keys[p], p.dtype, dt)  # a bucket's flat PARAMETER buffer has one dtype too (fp32 BatchNorm beside bf16 biases)
            # bucket_mb bounds the MESSAGE: with a wider reduction dtype that is what goes on the wire (cutting at the
            # gradients' own width made every fp32-reduced all-reduce 2x bucket_mb)
            esz = torch.empty((), dtype=dt).element_size()
            if reduce_dtype is not None and self.enabled:
                esz = max(esz, torch.empty((), dtype=reduce_dtype).element_size())
            nbytes = p.numel() * esz
            cur = open_buckets.get(k)
            if cur is not None and cur[1] + nbytes > cap:
                order.append((k, cur[0]))
                cur = None
            if cur is None:
                cur = [[], 0]
                open_buckets[k] = cur
            cur[0].append(p)
            cur[1] += nbytes
        for k, cur in open_buckets.items():
            if cur[0]:
                order.append((k, cur[0]))
        for (key, _pdt, dt), plist in order:
            self._make_bucket(plist, dt, key, flatten_params)
        if reduce_dtype is not None and self.enabled:
            # the all-reduce sums `rflat`; in "view" mode every p.grad stays a view of the UN-reduced `flat`, so an optimizer
            # that reads p.grad would step on local gradients without any error: only the copy-mode consumer (`reduced(b)`) is legal
            if any(b.flat.element_size() < torch.empty((), dtype=reduce_dtype).element_size() for b in self.buckets):
                assert grad_mode == "copy", "reduce_dtype wider than the gradients needs grad_mode='copy' (consumers read reduced(b))"
            for b in self.buckets:
                if b.flat.element_size() < torch.empty((), dtype=reduce_dtype).element_size():
                    b.rflat = torch.zeros(b.flat.numel(), dtype=reduce_dtype, device=self.device)

        if grad_mode == "copy":  # kernels that produce parameter gradients may write the bucket slots directly
            for b in self.buckets:
                for p, v in zip(b.params, b.views):
                    grad_sink.register(p, v)
        self._hooks = [p.register_post_accumulate_grad_hook(self._on_grad_ready) for p in params]
        self._launched = 0

    # ------------------------------------------------------------------------------------------
    def _make_bucket(self, params, dtype, key=None, flatten_params=False):
        # every parameter starts on a 16-byte boundary so flat-buffer kernels can use 16-byte accesses per tensor
        # (in BOTH flat buffers: with fp32 gradients of bf16 parameters the offset granule comes from the narrower element)
        esz = min(torch.empty((), dtype=dtype).element_size(), params[0].element_size() if flatten_params else 16)
        align = max(1, 16 // esz)
        offs, total = [], 0
        for p in params:
            offs.append(total)
            total += (p.numel() + align - 1) // align * align
        flat = torch.zeros(total, dtype=dtype, device=self.device)
        b = _Bucket(flat, params, key)
        b.offs = offs        # element offset of every parameter's slot (the same in the gradient and the parameter buffer)
        if flatten_params:
            b.pflat = torch.zeros(total, dtype=params[0].dtype, device=self.device)
        for p, off in zip(params, offs):
            n = p.numel()
            b.views.append(flat[off:off + n].view_as(p))
            if self.grad_mode == "view":
                p.grad = b.views[-1]  # autograd accumulates in place into the bucket
            if flatten_params:
                assert p.dtype == params[0].dtype
                pv = b.pflat[off:off + n].view_as(p)
                pv.copy_(p.data)
                p.data = pv
            self._bucket_of[p] = b
        self.buckets.append(b)

    def broadcast_module_state(self, src=0):
        """rank 0's parameters and buffers to everyone, in ONE flat message per dtype (DDP does this at wrap time)"""
        tensors = [p.data for p in self.module.parameters()] + [b.data for b in self.module.buffers()]
        by_dtype = {}
        for t in tensors:
            by_dtype.setdefault(t.dtype, []).append(t)
        for dt, ts in by_dtype.items():
            flat = torch.cat([t.reshape(-1) for t in ts])
            dist.broadcast(flat, src=src, group=self.group)
            off = 0
            for t in ts:
                t.copy_(flat[off:off + t.numel()].view_as(t))
                off += t.numel()
        # (`.data` writes move neither torch's version counter nor the optimizer epoch: weight packings made ahead of time are stale)
        from . import cot_layer_fused
        cot_layer_fused.invalidate_packs()

    # ------------------------------------------------------------------------------------------
    def _on_grad_ready(self, param):
        b = self._bucket_of[param]
        if b.pending <= 0 and (self.grad_mode == "copy" or (self.enabled and not self.defer_comm)):
            # a second backward pass before finish(): the bucket is already filled / its all-reduce in flight, so one micro-batch
            # would be overwritten (copy mode) or left out of the average.  The reference runs one backward per step
            # (train.py:264-274); a single process in view mode may accumulate (nothing was launched)
            raise RuntimeError("GradBucketReducer: a gradient arrived after its bucket was handed over -- call finish() once per "
                               "backward() (after an abandoned step, zero_grad() resets the buckets)")
        if self.grad_mode == "copy":
            b.fired.add(param)
            b.pending -= 1
            if b.pending == 0:
                self._fill(b)
                self._launch(b)
            return
        if param.grad is not None and param.grad.data_ptr() != self._expected_ptr(b, param):
            # someone replaced .grad (e.g. zero_grad(set_to_none=True)); fold it back into the bucket
            view = self._view(b, param)
            grad_sink.sync_producers()  # (a producer's side stream may still be writing the tensor that is read here)
            view.copy_(param.grad)
            param.grad = view
        b.pending -= 1
        if b.pending == 0:
            self._launch(b)

    def _fill(self, b):
        """copy mode: one multi-tensor copy of the gradients autograd produced into the bucket; zero the rest"""
        dst, src = [], []
        for p, v in zip(b.params, b.views):
            if p in b.fired and p.grad is not None:
                if not grad_sink.is_in_place(p, v):  # (the producing kernel may have written the slot itself: grad_sink)
                    dst.append(v)
                    src.append(p.grad)
            else:
                v.zero_()  # parameter unused this step
        if dst:
            grad_sink.sync_producers()  # (a gradient kernel on a side stream may still be writing what is copied here)
            torch._foreach_copy_(dst, src)
        for p in b.params:
            p.grad = None
            grad_sink.release(p)
        b.fired.clear()

    def _view(self, b, param):
        for p, v in zip(b.params, b.views):
            if p is param:
                return v
        raise KeyError("parameter not in bucket")

    def _expected_ptr(self, b, param):
        return self._view(b, param).data_ptr()

    def _launch(self, b, force=False):
        self._launched += 1
        if not self.enabled or (self.defer_comm and not force):
            return
        if self.on_gpu:
            b.ready_event = torch.cuda.Event()
            b.ready_event.record(torch.cuda.current_stream(self.device))
            with torch.cuda.stream(self.comm_stream):
                self.comm_stream.wait_event(b.ready_event)
                for ps in grad_sink.producer_streams():  # weight gradients written in place by a side stream (cot_layer_fused)
                    self.comm_stream.wait_stream(ps)
                if b.rflat is not None:
                    b.rflat.copy_(b.flat)  # (widening, on the communication stream: behind every producer, ahead of the collective)
                b.work = self._all_reduce(self.reduced(b))
        else:
            if b.rflat is not None:
                b.rflat.copy_(b.flat)
            b.work = self._all_reduce(self.reduced(b))

    @staticmethod
    def reduced(b):
        """the buffer that holds the bucket's all-reduced gradients after finish() / allreduce_all()"""
        return b.rflat if b.rflat is not None else b.flat

    def _all_reduce(self, flat):
        # Inside a HIP-graph capture the collective is issued synchronously (in stream order on the communication stream, no
        # Work object): `work.wait()` on a side stream of a capture segfaults in hipStreamEndCapture on ROCm 7.2 / torch 2.10
        # (scripts/probe_rccl_capture.py, profiles/r06_rccl_capture_probe.log), the stream-ordered form captures and replays.
        # The host does not block either way; finish() joins the communication stream.
        capturing = self.on_gpu and torch.cuda.is_current_stream_capturing()
        if self._avg_op:
            w = dist.all_reduce(flat, op=dist.ReduceOp.AVG, group=self.group, async_op=not capturing)
        else:
            flat.div_(self.world)  # pre-divide, then SUM (gloo has no AVG)
            w = dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=not capturing)
        return None if capturing else w

    # ------------------------------------------------------------------------------------------
    def finish(self):
        """call after backward, before optimizer.step(): the compute stream waits for every bucket's all-reduce.
        Buckets whose gradients never fired (unused parameters) are reduced here so ranks stay in lock-step."""
        for b in self.buckets:
            if self.grad_mode == "copy" and b.pending != 0:
                self._fill(b)  # some parameters were unused this step: their slots are zeroed
                b.pending = 0
                self._launch(b)
            elif b.work is None and self.enabled:
                self._launch(b)
        for b in self.buckets:
            if b.work is not None:
                if self.on_gpu:
                    with torch.cuda.stream(self.comm_stream):
                        b.work.wait()
                else:
                    b.work.wait()
                b.work = None
            b.pending = len(b.params)
        if self.on_gpu and self.enabled:
            torch.cuda.current_stream(self.device).wait_stream(self.comm_stream)

    def allreduce_all(self):
        """deferred-communication mode (HIP-graph replay: the hooks are not re-run): all-reduce every bucket now, on the
        communication stream, and make the compute stream wait for the result.  No host synchronisation."""
        if not self.enabled:
            return
        for b in self.buckets:
            self._launch(b, force=True)
        for b in self.buckets:
            if b.work is not None:
                if self.on_gpu:
                    with torch.cuda.stream(self.comm_stream):
                        b.work.wait()
                else:
                    b.work.wait()
                b.work = None
        if self.on_gpu:
            torch.cuda.current_stream(self.device).wait_stream(self.comm_stream)

    def zero_grad(self):
        """zero the flat buckets (keeps the .grad views alive; use instead of optimizer.zero_grad(set_to_none=True)).
        In copy mode nothing accumulates, so this only drops stale .grad tensors.  Also the recovery path after a step that was
        abandoned between backward() and finish() (exception, skipped NaN step): in-flight all-reduces are waited for and
        every bucket's bookkeeping is reset, so the next backward() starts a fresh step."""
        for b in self.buckets:
            if b.work is not None:
                if self.on_gpu:
                    with torch.cuda.stream(self.comm_stream):
                        b.work.wait()
                else:
                    b.work.wait()
                b.work = None
            b.fired.clear()
            b.pending = len(b.params)
        if self.on_gpu and self.enabled:
            torch.cuda.current_stream(self.device).wait_stream(self.comm_stream)
        if self.grad_mode == "copy":
            for b in self.buckets:
                for p in b.params:
                    p.grad = None
                    grad_sink.release(p)
            return
        for b in self.buckets:
            b.flat.zero_()
            for p in b.params:
                if p.grad is None or p.grad.data_ptr() != self._expected_ptr(b, p):
                    p.grad = self._view(b, p)

    def remove(self):
        """detach from the module: hooks off, gradient-sink entries dropped (they hold strong references to every parameter
        and bucket view -- a reducer that is merely garbage-collected would keep its flat buffers alive on the GPU)"""
        for h in self._hooks:
            h.remove()
        self._hooks = []
        if self.grad_mode == "copy":
            for b in self.buckets:
                for p, v in zip(b.params, b.views):
                    grad_sink.unregister(p, v)  # (only this reducer's own entries: a successor may have re-registered p)

    def __del__(self):
        try:
            self.remove()
        except Exception:  # interpreter shutdown
            pass


def distribute_bn(module, process_group=None, reduce=True):
    """Average (reduce=True) or broadcast-from-rank-0 every BatchNorm running_mean / running_var across ranks in
    ONE flat collective (the reference issues one tiny all-reduce per buffer, utils/distributed.py:57-67)."""
    if not (dist.is_available() and dist.is_initialized()):
        return
    world = dist.get_world_size(process_group)
    if world == 1:
        return
    bufs = [b for n, b in module.named_buffers() if ("running_mean" in n) or ("running_var" in n)]
    if not bufs:
        return
    flat = torch.cat([b.reshape(-1).float() for b in bufs])
    if reduce:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=process_group)
        flat /= float(world)
    else:
        dist.broadcast(flat, src=0, group=process_group)
    off = 0
    for b in bufs:
        b.copy_(flat[off:off + b.numel()].view_as(b))
        off += b.numel()

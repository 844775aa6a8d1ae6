"""CotLayer / CoXtLayer .forward -- and the Bottleneck around them -- as ONE autograd node (default on; COT_FUSED_LAYER=0 opts out).

Why: at the reference's batch (80 / GPU) the round-1 CoTNet-50 step on MI355X was bound by the host, not the device
(~3100 launches, ~38 ms of Python/dispatch per step against 36 ms of kernels).  One CotLayer is ~14 autograd
nodes forward and as many backward, plus the engine's gradient-accumulation adds wherever a tensor has several consumers
(x feeds key_embed, embed and conv1x1: two adds of C*H*W; k feeds embed and the radix tail: one more).  This module
evaluates the same function (models/cotnet.py:79-104) by calling the library's kernels back to back through the C ABI:

    forward   conv3x3g -> bn+relu -> conv1x1 on [x | k] (no cat) -> bn+relu -> conv1x1(+bias) -> GroupNorm
              -> conv1x1 -> bn -> aggregation -> bn+silu -> radix gap (channel-major) -> se branch as two 1x1
              convolutions over the batch axis with its BatchNorm+ReLU between them -> pair softmax + radix mix
    backward  the mirror image; the three contributions to dx and the two to dk are summed inside the data-gradient
              kernels (`accumulate`), not by separate add kernels.

Same parameters, buffers (running statistics, num_batches_tracked) and state_dict as the module it is applied to;
eligible in training mode for bf16 NCHW tensors with dim % 64 == 0 (every CoTNet stage; CoXtLayer -- CoTNeXt's grouped
variant, models/cotnet.py:106-178 -- with dim % 32 == 0: its grouped 1x1 convolutions run on cot_conv1x1g_*, [x, k] is
interleaved once and the group -> batch fold is a view); anything else takes the module's ordinary forward.  Verified against the unfused reference formula through the host-emulated kernels
(tests/test_kernels_emulated.py) and on the GPU against an fp32 truth by tests/test_fused_layer_gpu.py; round 2 on the
MI355X: ~1290 launches and 19.1 ms per step with every kernel from cotnet_amd/csrc.
"""
import ctypes
import functools
import threading
import os
import weakref

import torch
from torch import nn
from torch.autograd import Function

from . import _lib, grad_sink

ENABLED = os.environ.get("COT_FUSED_LAYER", "1") != "0"  # default on; COT_FUSED_LAYER=0 = one autograd node per op
_DEVICE_ONLY = True  # tests drive the node on CPU tensors through the host-emulated kernels
BF16 = _lib.COT_BF16


def _p(t):
    # a plain int is accepted for a c_void_p parameter and skips building a ctypes object per argument (~1000 per step)
    return t.data_ptr() if t is not None else None


_TLS = threading.local()  # .st: the compute stream's handle for the node invocation running on this thread


def _stream():
    if not _DEVICE_ONLY:
        return None
    st = getattr(_TLS, "st", None)
    return st if st is not None else ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _one_stream_query(fn):
    """torch.cuda.current_stream() costs ~9 us and a node makes ~100 launches: ask once per forward / backward of a node (the
    current stream cannot change inside one; forward and backward run on different threads, hence thread-local).  Measured:
    1.3 ms of a step's ~12 ms of host time (gpurun_out/r3_cpu_profile.log)"""
    @functools.wraps(fn)
    def wrapped(*a, **k):
        if not _DEVICE_ONLY or getattr(_TLS, "st", None) is not None:
            return fn(*a, **k)
        _TLS.st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        try:
            return fn(*a, **k)
        finally:
            _TLS.st = None
    return wrapped


def _ck(rc, what):
    if rc:
        _lib.check(rc, what)


# ---- weight gradients on a side stream.  A parameter gradient is needed only when the step's backward is over, while the
# data-gradient chain it hangs off is the critical path: 100 weight-gradient launches (+ their split reduces) per CoTNet-50
# step, ~3 ms of device time in which 256-workgroup GEMM slices and their launch ramps sit between bandwidth-bound
# BatchNorm / data-gradient kernels that never need all the matrix pipes.  So every weight gradient of a node's backward is
# issued on ONE extra HIP stream per device: it waits (event) for what the compute stream has produced so far, runs beside
# the rest of the node's backward, and the compute stream joins it (event) before the node returns -- i.e. before autograd
# hands the gradients to anybody.  Rules that keep this race-free:
#   * a tensor read by a side-stream launch is never written again inside the node (no buffer reuse for them below) and is
#     kept referenced until the join (the caching allocator could otherwise hand its memory to a later compute-stream tensor);
#   * the side stream is in-order, so its launches share ONE persistent workspace of their own (partial sums are consumed by
#     the same launch's reduce); the compute stream's workspace is never touched by it.
# COT_WGRAD_STREAM=0 (or cot_layer_fused.SIDE_WGRAD = False) issues everything on the compute stream as before (A/B, tests).
#
# COT_WGRAD_LAZY=1 (default): the node's weight-gradient launches are QUEUED and issued together when its backward ends --
# one event pair for the node instead of one per launch (ten per Bottleneck: each pair is ~8 us of host time and a barrier
# packet on both queues) -- and the compute stream does not join them there: they run beside the NEXT node's backward.  The
# compute stream joins the side stream (a) at the next node's flush, which is also when the tensors the previous node's
# launches read are released, and (b) when the backward pass ends (autograd engine callback), i.e. before an optimizer or
# anybody else on the compute stream can see the gradients; a gradient bucket that launches its all-reduce mid-backward makes
# its communication stream wait for the side stream (grad_sink.producer_streams).  The un-joined form is taken ONLY by nodes
# whose parameter gradients are all flat-bucket aliases that autograd merely adopts (_Side.adopted_only: FlatSGD / the
# copy-mode reducer, the measured configuration); any node that hands autograd an ordinary gradient tensor, runs under
# create_graph or sees a foreign hook joins its side stream before it returns, exactly like COT_WGRAD_LAZY=0.
SIDE_WGRAD = os.environ.get("COT_WGRAD_STREAM", "1") != "0"
LAZY_WGRAD = os.environ.get("COT_WGRAD_LAZY", "1") != "0"
_SIDE_STREAMS = {}  # device index -> [torch.cuda.Stream, workspace tensor]
_SIDE_PENDING = {}  # device index -> {"keep": tensors read by launches the compute stream has not joined yet}


def _join_pending(dev_index):
    """the compute stream waits for everything issued on the side stream so far; the tensors those launches read are released"""
    ent, pend = _SIDE_STREAMS.get(dev_index), _SIDE_PENDING.get(dev_index)
    if ent is None or pend is None:
        return
    if pend["keep"]:
        cur = torch.cuda.current_stream(torch.device("cuda", dev_index))
        cur.wait_stream(ent[0])
        main = pend.get("main")  # the stream the node's backward ran on: the caller may sit on another one by now (backward
        if main is not None and main != cur:  # under a different ambient stream, re-entrant backward)
            main.wait_stream(ent[0])
        pend["keep"] = []


class _Side:
    """the side stream for one node's backward (or a pass-through onto the compute stream when switched off / on CPU tensors)"""
    __slots__ = ("on", "main", "stream", "st", "ws", "keep", "lazy", "queue", "dev", "fresh0", "params")

    def __init__(self, dev, ws_bytes, main_ws, params=()):
        self.on = SIDE_WGRAD and _DEVICE_ONLY and dev.type == "cuda"
        self.lazy = self.on and LAZY_WGRAD
        self.keep, self.queue, self.dev = [], [], dev
        self.fresh0, self.params = grad_sink.fresh_count(), params
        if not self.on:
            self.st, self.ws = _stream(), main_ws
            return
        ent = _SIDE_STREAMS.get(dev.index)
        if ent is None:
            ent = _SIDE_STREAMS[dev.index] = [torch.cuda.Stream(device=dev), None]
        self.main = torch.cuda.current_stream(dev)
        self.stream = ent[0]
        if ent[1] is None or ent[1].numel() < ws_bytes:
            with torch.cuda.stream(self.stream):  # (owned by the side stream's pool; grown, never shrunk)
                ent[1] = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
        self.ws = ent[1]
        self.st = ctypes.c_void_p(self.stream.cuda_stream)
        if self.lazy and dev.index not in _SIDE_PENDING:
            _SIDE_PENDING[dev.index] = {"keep": []}
            grad_sink.register_producer_stream(self.stream, lambda i=dev.index: _join_pending(i))

    def run(self, fn, *tensors):
        """fn(stream handle) launches ONE weight gradient that reads `tensors` (all of them final when this is called)"""
        if self.lazy:
            self.queue.append(fn)
            self.keep.extend(tensors)
        else:
            fn(self.ready(*tensors))

    def ready(self, *tensors):
        """call right before a side-stream launch that reads `tensors`: everything the compute stream has issued so far is
        ordered before it, and the tensors stay alive until join()"""
        if self.on:
            self.stream.wait_stream(self.main)
            self.keep.extend(tensors)
        return self.st

    def adopted_only(self):
        """True when nobody can READ this node's parameter gradients before the backward pass ends: every one of them is an
        alias of its flat-bucket slot (grad_sink: autograd adopts it, and the copy-mode reducer that owns the slot makes its
        communication stream wait for the side stream), no graph is being recorded (create_graph=True makes AccumulateGrad
        clone what it is handed) and no foreign tensor / post-accumulate hook sits on a parameter.  An ORDINARY gradient
        tensor is accumulated by autograd on the compute stream right after the node returns (`p.grad += g` when a gradient
        exists already: a second backward without zero_grad, micro-batches, a view-mode reducer, a parameter used twice) --
        while the side stream may still be writing it.  Only the first kind of node may leave its weight
        gradients un-joined."""
        if grad_sink.fresh_count() != self.fresh0 or torch.is_grad_enabled():
            return False
        for p in self.params:
            if p._backward_hooks:
                return False
            h = p._post_accumulate_grad_hooks
            if h is not None and len(h) > 1:  # (one = the reducer that registered the sink)
                return False
        return True

    def join(self):
        if self.lazy:
            pend = _SIDE_PENDING[self.dev.index]
            if pend["keep"]:  # the previous node's launches: long done by now; their inputs may go
                self.main.wait_stream(self.stream)
            pend["keep"], pend["main"] = self.keep, self.main
            self.stream.wait_stream(self.main)
            for fn in self.queue:
                fn(self.st)
            self.queue, self.keep = [], []
            if not self.adopted_only():  # somebody may read the gradients as soon as the node returns: join here
                self.main.wait_stream(self.stream)
                pend["keep"] = []
                return
            # join when the backward pass ends.  Queued by EVERY flush (the first callback to run joins, the others find nothing
            # pending): a flag "already queued" would survive a backward pass that died half-way and silence all later ones
            try:
                torch.autograd.Variable._execution_engine.queue_callback(lambda i=self.dev.index: _join_pending(i))
            except RuntimeError:  # (not inside the engine's backward: join now)
                _join_pending(self.dev.index)
        elif self.on:
            self.main.wait_stream(self.stream)
            self.keep.clear()


GN_FUSED = os.environ.get("COT_GN_FUSED", "1") != "0"  # GroupNorm-9 statistics in embed[3]'s epilogue + normalisation in the aggregation's prologue
_GN_OK = _lib.register_cache({})


def _gn_fused_ok(L, Ch, HW, W):
    k = (Ch, HW, W)
    v = _GN_OK.get(k)
    if v is None:
        v = _GN_OK[k] = bool(L.cot_gn9_fused_covers(Ch, Ch, 0, HW, W))
    return v


# BatchNorm + SiLU of the aggregation's output folded into the radix tail (round 6; VERDICT r5 J1 / next #2b; models/cotnet.py:89-104):
# a statistics pass over `a`, then the pooling and the mix kernels normalise + activate as they load (cot_radix_*_bn) -- y = silu(bn(a))
# is never written; backward: two launches (reduce, apply) instead of four, the BatchNorm's channel sums out of the reduce kernel's
# plane sums.  COT_BN_TAIL=0 = the separate cot_bn_act_* kernels as before.
BN_TAIL = os.environ.get("COT_BN_TAIL", "1") != "0"


# ... and, opt-in (COT_AGG_ROWSTATS=1), its statistics out of the aggregation's own epilogue (cot_agg_forward_rowstats +
# cot_bn_rowstats_finalize) where the LDS forward kernel serves the geometry: no statistics pass over `a` at all.  Built, parity-green and
# measured: the epilogue costs the aggregation +5.7 / +3.0 / +1.7 / -0.8 us at 56 / 28 / 14 / 7 pixels and its finalize 6.4-7.4 us, against
# 10.3-14.9 us for cot_bn_batch_stats -- 37.1 -> 35.3, 26.6 -> 26.1, 20.5 -> 18.3, 20.7 -> 15.4 us per layer, and the step does not move
# (13.56 vs 13.56 ms, alternating; profiles/r06_rowstats_kernels.log, r06_rowstats_ab.log): default off.
AGG_ROWSTATS = os.environ.get("COT_AGG_ROWSTATS", "0") == "1"
_ROWSTATS_OK = _lib.register_cache({})


def _agg_fwd_stats(L, v, w_or_logits, a, gn, gn_mean, gn_rstd, geom, bn, stats, N, C, H, W):
    """a = aggregation(v, w) and what bn's batch statistics of a are made of.  -> True: mean / rstd are in stats[:C] / stats[C:2C] and
    the running statistics updated; False: stats[2C:] holds chunk sums that the pooling kernel's prologue finalizes (_tail_gap).  gn
    given: w_or_logits holds the raw logits and the GroupNorm is applied in the aggregation's prologue (cot_agg_gn9_forward's contract)."""
    st = _stream()
    key = (geom.N, geom.C, H, W, geom.wC, gn is not None)
    if AGG_ROWSTATS and _ROWSTATS_OK.get(key, True):
        rows = torch.empty(int(L.cot_agg_rowstats_floats(N, C, H)), dtype=torch.float32, device=a.device)
        rc = L.cot_agg_forward_rowstats(_p(v), _p(w_or_logits), _p(a), _p(rows), _p(gn_mean) if gn is not None else None,
                                        _p(gn_rstd) if gn is not None else None, _p(gn.weight) if gn is not None else None,
                                        _p(gn.bias) if gn is not None else None, gn.num_groups if gn is not None else 0,
                                        ctypes.byref(geom), BF16, st)
        if rc == 0:
            _ck(L.cot_bn_rowstats_finalize(_p(rows), _p(stats), _p(stats[C:]), _p(bn.running_mean), _p(bn.running_var),
                                           _p(bn.num_batches_tracked), N, C, H, W, float(bn.eps), float(bn.momentum), st),
                "cot_bn_rowstats_finalize")
            return True
        if rc != _lib.COT_ERR_UNSUPPORTED:
            _ck(rc, "cot_agg_forward_rowstats")
        _ROWSTATS_OK[key] = False  # (geometry off the LDS forward kernel: the plain forward + a statistics pass, from now on without asking)
    if gn is not None:
        _ck(L.cot_agg_gn9_forward(_p(v), _p(w_or_logits), _p(gn_mean), _p(gn_rstd), _p(gn.weight), _p(gn.bias), gn.num_groups, _p(a),
                                  ctypes.byref(geom), BF16, st), "cot_agg_gn9_forward")
    else:
        _ck(L.cot_agg_forward(_p(v), _p(w_or_logits), _p(a), ctypes.byref(geom), BF16, _lib.COT_NCHW, st), "cot_agg_forward")
    _ck(L.cot_bn_stats_sums(_p(a), _p(stats[2 * C:]), N, C, H * W, BF16, st), "cot_bn_stats_sums")
    return False


def _tail_gap(L, a, k, gapT, bn, stats, final, N, C, HW, lay):
    """gapT = mean_hw(silu(bn(a)) + k); final False: stats[2C:] holds cot_bn_stats_sums' chunk sums and this launch finalizes them"""
    _ck(L.cot_radix_gap_t_bn(_p(a), _p(k), _p(gapT), _p(bn.weight), _p(bn.bias), _p(stats), _p(stats[C:]), _p(bn.running_mean),
                             _p(bn.running_var), _p(bn.num_batches_tracked), None if final else _p(stats[2 * C:]), N, C, HW, float(bn.eps),
                             float(bn.momentum), lay, BF16, _stream()), "cot_radix_gap_t_bn")

_SIZES = _lib.register_cache({})  # (N, C, H, W, A) -> (workspace bytes, bn workspace floats for C, C/2 and the se branch's A channels)
_MASKS = {}


def _sizes(L, N, C, H, W, A, G, grouped=False):
    k = (N, C, H, W, A, G, grouped)
    v = _SIZES.get(k)
    if v is None:
        HW = H * W
        if grouped:  # CoXtLayer: the three 1x1 convolutions of the layer are grouped (groups = 2): general kernels' workspace
            ws = max(int(L.cot_conv3x3g_workspace(N, C, C, G, H, W)), int(L.cot_conv3x3g_workspace(N, C, C, _conv3x3_ws_groups(C, G), H, W)),
                     int(L.cot_convg_workspace(N, 2 * C, C // 2, 2, H, W, 1)),
                     int(L.cot_convg_workspace(N, C // 2, 9 * C // 8, 2, H, W, 1)), int(L.cot_convg_workspace(N, C, C, 2, H, W, 1)),
                     int(L.cot_conv1x1_workspace(1, C, A, N, 1)), int(L.cot_conv1x1_workspace(1, A, 2 * C, N, 1)))
        else:
            ws = max(int(L.cot_conv3x3g_workspace(N, C, C, G, H, W)), int(L.cot_conv1x1_workspace(N, 2 * C, C // 2, HW, 0)),
                     int(L.cot_conv1x1_workspace(N, C // 2, 9 * C // 8, HW, 1)), int(L.cot_conv1x1_workspace(N, C, C, HW, 0)),
                     int(L.cot_conv1x1_workspace(1, C, A, N, 1)), int(L.cot_conv1x1_workspace(1, A, 2 * C, N, 1)))
        v = _SIZES[k] = (ws, int(L.cot_bn_act_workspace(N, C)), int(L.cot_bn_act_workspace(N, C // 2)),
                         int(L.cot_bn_act_workspace(1, A)))
    return v


def _masks(L, H, W, device):
    k = (H, W, str(device))
    m = _MASKS.get(k)
    if m is None:
        m = torch.empty(int(L.cot_conv3x3g_masks_bytes(H, W)), dtype=torch.uint8, device=device)
        _ck(L.cot_conv3x3g_masks(_p(m), H, W, _stream()), "cot_conv3x3g_masks")
        _MASKS[k] = m
    return m


# ---- grouped 3x3 weights packed ahead of time (round 5).  The LDS kernels run on a re-ordered copy of the weights;
# the ordinary entry points make it per call (two 5 us launches per layer and step on the compute stream: 32 launches, 0.16 ms).  A
# layer's weights change once per optimizer step and each packing is used once per step, so nothing is saved in work -- but the packing
# need not sit on the critical path: right after the flat SGD kernels (FlatSGD.step -> after_optimizer_step) both packings of every
# layer seen in the last forward are made on the weight-gradient side stream, into persistent buffers, and the next step's forward /
# backward run `cot_conv3x3g_*_packed`.  Validity = (storage pointer, torch's version counter, the optimizer epoch below -- the flat
# SGD kernel writes through raw pointers, which torch's counter does not see --, geometry).  Never inside a graph capture (a replay
# would keep reading the packing of the capture): captured steps pack inline as before.  (Round 6 tried the scheme INSIDE a whole-step
# capture -- packings re-made behind the captured SGD kernels, read by the next replay; loss trajectory identical to eager -- and
# measured nothing: 13.82 / 13.80 ms with / without, profiles/r06_prepack_in_capture_ab.log; the inline packing launches are not on
# a replayed step's critical path.  Not kept.)  COT_PREPACK=0 opts out.
PREPACK = os.environ.get("COT_PREPACK", "1") != "0"
PARAM_EPOCH = [0]
_PACKS = weakref.WeakKeyDictionary()   # nn.Conv2d -> {"geom": (N, C, G, H, W), 0: (key, buffer), 1: (key, buffer)}
_PACK_EVENT = {}                       # device index -> event the compute stream has to wait for before its first packed launch


def _capturing():
    return _DEVICE_ONLY and torch.cuda.is_available() and torch.cuda.is_current_stream_capturing()


def _pack_for(L, conv, mode, N, C, G, H, W):
    """-> the valid packing of conv.weight for this call, or None (the caller then packs inline through the ordinary entry point).
    Also remembers the geometry for after_optimizer_step."""
    if not (PREPACK and _DEVICE_ONLY):
        return None
    e = _PACKS.get(conv)
    if e is None:
        e = _PACKS[conv] = {}
    if _capturing():
        e.pop("geom", None)
        return None
    wt = conv.weight
    e["geom"] = (N, C, G, H, W)
    ent = e.get(mode)
    if ent is not None and ent[0] == (wt.data_ptr(), wt._version, PARAM_EPOCH[0], N, C, G, H, W):
        ev = _PACK_EVENT.pop(wt.device.index, None)
        if ev is not None:
            torch.cuda.current_stream(wt.device).wait_event(ev)
        return ent[1]
    return None


def invalidate_packs():
    """Every weight packing made ahead of time is stale from here on: the next packed call packs inline again.

    A packing is keyed on (storage pointer, torch's version counter, PARAM_EPOCH, geometry).  Two kinds of write move neither
    counter by themselves and MUST call this: (1) replaying a HIP graph that captured the optimizer's kernels (the flat SGD kernel
    writes through raw pointers; FlatSGD.step() bumps the epoch only when it runs in Python), (2) raw `.data` writes -- `p.data.
    copy_()`, a broadcast of the module's state, an EMA swap-in.  GradBucketReducer.broadcast_module_state and bench.py's replay
    loop call it; code that writes parameters behind torch's back has to as well."""
    PARAM_EPOCH[0] += 1
    _PACK_EVENT.clear()


def after_optimizer_step(device):
    """called by FlatSGD.step() once its kernels are issued: the parameters have changed (PARAM_EPOCH), and the 3x3 weights of the
    layers the last forward ran are packed for the next step on the side stream"""
    PARAM_EPOCH[0] += 1
    if not (PREPACK and ENABLED and SIDE_WGRAD and _DEVICE_ONLY and device.type == "cuda") or _capturing():
        return
    ent = _SIDE_STREAMS.get(device.index)
    if ent is None or not _PACKS:
        return
    L = _lib.lib()
    side, cur = ent[0], torch.cuda.current_stream(device)
    side.wait_stream(cur)  # behind the SGD kernels (and with them behind the whole backward pass that read the old packings)
    st = ctypes.c_void_p(side.cuda_stream)
    done = False
    for conv, e in list(_PACKS.items()):
        g = e.get("geom")
        wt = conv.weight
        if g is None or wt.device != device or wt.dtype != torch.bfloat16:
            continue
        N, C, G, H, W = g
        key = (wt.data_ptr(), wt._version, PARAM_EPOCH[0], N, C, G, H, W)
        for mode in (0, 1):
            old = e.get(mode)
            buf = old[1] if old is not None else None
            if buf is None:
                nb = int(L.cot_conv3x3g_packed_bytes(C, C, G))
                if nb <= 0:
                    break
                buf = torch.empty(nb, dtype=torch.uint8, device=device)
            if L.cot_conv3x3g_pack(_p(wt), _p(buf), mode, N, C, C, G, H, W, BF16, st) == 0:
                e[mode] = (key, buf)
                done = True
            else:
                e.pop(mode, None)  # (geometry off the LDS kernels: they gather the weights in place)
    if done:
        ev = torch.cuda.Event()
        ev.record(side)
        _PACK_EVENT[device.index] = ev


# ---- groups of 12 channels (CoXtLayer(96).key_embed: 96 channels in 8 groups, models/cotnet.py:113-117): the LDS-pipelined 3x3 kernels take
# groups from 16 channels on, the general kernel (conv_gen.hip) runs this layer at 169 us per launch at 56 x 56, B = 64.  Two neighbouring
# groups ARE one group of 24 channels with a block-diagonal weight (the off-diagonal blocks zero), which the LDS kernels take: a copy of the
# 12 x 12 x 9 blocks onto the diagonal of a persistent zero-filled [C, 24, 3, 3] buffer per step (one small launch) instead.  The weight
# gradient keeps the module's grouping (its kernel merges neighbours itself).  COT_MERGE12=0 opts out.
MERGE12 = os.environ.get("COT_MERGE12", "1") != "0"
_MERGED = weakref.WeakKeyDictionary()  # nn.Conv2d -> [key, merged weight]


def _merge12(C, G):
    return MERGE12 and G % 2 == 0 and C == 12 * G


def _merged_weight(conv, C, G, refresh):
    w = conv.weight
    ent = _MERGED.get(conv)
    if ent is None or ent[1].device != w.device or ent[1].dtype != w.dtype:
        ent = _MERGED[conv] = [None, torch.zeros((C, 24, 3, 3), dtype=w.dtype, device=w.device)]
    key = (w.data_ptr(), w._version, PARAM_EPOCH[0])
    if refresh or ent[0] != key:  # (forward: always -- a replayed graph moves the weights behind every counter; backward: the forward's copy)
        G2 = G // 2
        ent[1].view(G2, 2, 12, 2, 12, 9).diagonal(dim1=1, dim2=3).copy_(w.detach().view(G2, 2, 12, 12, 9).permute(0, 2, 3, 4, 1))
        ent[0] = key
    return ent[1]


def _conv3x3_ws_groups(C, G):
    """the group count the 3x3 forward / data gradient are launched with (workspace sizing)"""
    return G // 2 if _merge12(C, G) else G


def _conv3x3_fwd(L, conv, x, y, masks, ws, N, C, G, H, W):
    if _merge12(C, G):
        wm = _merged_weight(conv, C, G, True)
        _ck(L.cot_conv3x3g_forward(_p(x), _p(wm), _p(y), _p(masks), _p(ws), N, C, C, G // 2, H, W, BF16, _stream()), "cot_conv3x3g_forward")
        return
    pk = _pack_for(L, conv, 0, N, C, G, H, W)
    if pk is not None:
        _ck(L.cot_conv3x3g_forward_packed(_p(x), _p(pk), _p(y), N, C, C, G, H, W, BF16, _stream()), "cot_conv3x3g_forward_packed")
    else:
        _ck(L.cot_conv3x3g_forward(_p(x), _p(conv.weight), _p(y), _p(masks), _p(ws), N, C, C, G, H, W, BF16, _stream()), "cot_conv3x3g_forward")


def _conv3x3_dgrad(L, conv, gy, gx, accumulate, masks, ws, N, C, G, H, W):
    if _merge12(C, G):
        wm = _merged_weight(conv, C, G, False)
        _ck(L.cot_conv3x3g_backward_data(_p(gy), _p(wm), _p(gx), accumulate, _p(masks), _p(ws), N, C, C, G // 2, H, W, BF16, _stream()),
            "cot_conv3x3g_backward_data")
        return
    pk = _pack_for(L, conv, 1, N, C, G, H, W)
    if pk is not None:
        _ck(L.cot_conv3x3g_backward_data_packed(_p(gy), _p(pk), _p(gx), accumulate, N, C, C, G, H, W, BF16, _stream()),
            "cot_conv3x3g_backward_data_packed")
    else:
        _ck(L.cot_conv3x3g_backward_data(_p(gy), _p(conv.weight), _p(gx), accumulate, _p(masks), _p(ws), N, C, C, G, H, W, BF16, _stream()),
            "cot_conv3x3g_backward_data")


class _Plan:
    """per-layer handles resolved once (nn.Sequential indexing and parameter walks cost more than a kernel launch)"""
    __slots__ = ("ke0", "ke1", "em0", "em1", "em3", "gn", "cv0", "cv1", "bn", "se0", "sebn", "se3", "params",
                 "static_ok", "grouped")

    def __init__(self, layer):
        ke, em, cv = layer.key_embed, layer.embed, layer.conv1x1
        # CoXtLayer (models/cotnet.py:106-178): the 1x1 convolutions are grouped (dw_group = 2), the key embedding has 8
        # groups, [x, k] is channel-INTERLEAVED and the two groups are folded into the batch for the aggregation
        g1 = int(getattr(layer, "dw_group", 1))
        self.grouped = g1 == 2
        self.ke0, self.ke1, self.em0, self.em1, self.em3, self.gn = ke[0], ke[1], em[0], em[1], em[3], em[4]
        se = layer.se
        self.cv0, self.cv1, self.bn, self.se0, self.sebn, self.se3 = cv[0], cv[1], layer.bn, se[0], se[1], se[3]
        self.params = [ke[0].weight, ke[1].weight, ke[1].bias, em[0].weight, em[1].weight, em[1].bias, em[3].weight,
                       em[3].bias, em[4].weight, em[4].bias, cv[0].weight, cv[1].weight, cv[1].bias, layer.bn.weight,
                       layer.bn.bias, se[0].weight, se[0].bias, se[1].weight, se[1].bias, se[3].weight, se[3].bias]
        C = layer.dim
        self.static_ok = (
            g1 in (1, 2) and C % (16 * g1 if self.grouped else 64) == 0 and layer.kernel_size == 3
            and isinstance(layer.act, nn.SiLU) and layer.radix == 2
            and _conv_ok(ke[0], 3) and ke[0].bias is None and (self.grouped or (C // ke[0].groups) % 8 == 0)
            and isinstance(ke[2], nn.ReLU) and _conv_ok(em[0], 1, g1) and em[0].bias is None
            and isinstance(em[2], nn.ReLU) and _conv_ok(em[3], 1, g1) and em[3].bias is not None
            and isinstance(em[4], nn.GroupNorm) and em[4].num_groups * 9 == em[4].num_channels and em[4].affine
            and _conv_ok(cv[0], 1, g1) and cv[0].bias is None
            and len(se) == 4 and _conv_ok(se[0], 1, 1) and se[0].bias is not None and isinstance(se[2], nn.ReLU)
            and _conv_ok(se[3], 1, 1) and se[3].bias is not None and se[0].out_channels % 8 == 0
            and se[3].out_channels == 2 * C
            and all(_bn_static_ok(b) for b in (ke[1], em[1], cv[1], layer.bn, se[1])))


_PLANS = weakref.WeakKeyDictionary()  # module -> plan (kept out of the module's __dict__: deepcopy / pickling / state_dict
_BLOCK_PLANS = weakref.WeakKeyDictionary()  # never see it, and a copied module gets a plan of its own)


def _plan(layer):
    p = _PLANS.get(layer)
    if p is None:
        p = _PLANS[layer] = _Plan(layer)
    return p


def _new_guarded(N, C, H, W, dtype, dev):
    """[N, C, H, W] tensor with W + 1 (rounded up to 8) elements of the same allocation before and behind it: the grouped 3x3
    weight gradient's LDS-staged kernel copies x at pixel + tap offset in whole 16-byte pieces
    (cot_conv3x3g_backward_weight_guarded; what lies in the margins never reaches a sum)"""
    lead = (W + 1 + 7) // 8 * 8
    n = N * C * H * W
    flat = torch.empty(n + 2 * lead, dtype=dtype, device=dev)
    return flat[lead:lead + n].view(N, C, H, W)


def _guard_elems(t):
    """elements of t's own allocation before its first and behind its last element (0 for a tensor that fills its storage)"""
    if not t.is_contiguous():
        return 0
    total = t.untyped_storage().nbytes() // t.element_size()
    return max(0, min(t.storage_offset(), total - t.storage_offset() - t.numel()))


RELU_MASK = os.environ.get("COT_BN_RELU_MASK", "1") != "0"  # bn3 + residual + ReLU: backward reads a 1-bit sign mask, not y
# identity-shortcut blocks: the residual's gradient gout * [block output > 0] is not written by bn3's backward and read back by conv1's
# data gradient (`accumulate`) but formed in that data gradient's epilogue from gout and the sign mask (cot_conv1x1_backward_data_relu_res):
# one tensor write and most of one read per block less, one backward buffer less.  Opt-in (COT_RES_FOLD=1): bit-identical results, and the
# step does not move (13.556 vs 13.552 ms, alternating; profiles/r06_res_fold_ab.log) -- bn3's backward is not where these blocks wait.
RES_FOLD = os.environ.get("COT_RES_FOLD", "0") == "1"
_RES_FOLD_OK = _lib.register_cache({})


def _res_fold_ok(L, N, Ci, Co, HW):
    if not RES_FOLD:
        return False
    k = (N, Ci, Co, HW)
    v = _RES_FOLD_OK.get(k)
    if v is None:
        v = _RES_FOLD_OK[k] = bool(L.cot_conv1x1_backward_data_relu_res_covers(N, Ci, Co, HW, BF16))
    return v

_MASK_BYTES = _lib.register_cache({})


def _relu_mask(L, N, C, HW, dev):
    """uint8 tensor for the ReLU sign mask of a bn + residual + ReLU over [N, C, HW] (cot_bn_act_*_mask), or None when the
    library does not take one for this geometry (then the backward reads the saved output as before)"""
    if not RELU_MASK:
        return None
    k = (N, C, HW)
    nb = _MASK_BYTES.get(k)
    if nb is None:
        nb = _MASK_BYTES[k] = int(L.cot_bn_relu_mask_bytes(N, C, HW, BF16))
    return torch.empty(nb, dtype=torch.uint8, device=dev) if nb > 0 else None


def _bn_fwd(L, x, y, bn, stats, nws_off, N, C, HW, act, residual=None, ps=None, mask=None):
    """stats: fp32 [2*C + workspace] -> mean = stats[:C], rstd = stats[C:2C].  ps: per-sample scale of the normalised branch
    (stochastic depth: 0 or 1 / keep, fp32 [N]) or None.  mask: `_relu_mask` tensor to fill (act = ReLU with a residual)"""
    if mask is not None:
        _ck(L.cot_bn_act_forward_mask(_p(x), _p(residual), _p(y), _p(mask), _p(bn.weight), _p(bn.bias), _p(stats), _p(stats[C:]),
                                      _p(bn.running_mean), _p(bn.running_var), _p(bn.num_batches_tracked), _p(stats[nws_off:]),
                                      _p(ps), N, C, HW, float(bn.eps), float(bn.momentum), act, BF16, _stream()),
            "cot_bn_act_forward_mask")
        return
    if ps is not None:
        _ck(L.cot_bn_act_forward_ps(_p(x), _p(residual), _p(y), _p(bn.weight), _p(bn.bias), _p(stats), _p(stats[C:]),
                                    _p(bn.running_mean), _p(bn.running_var), _p(bn.num_batches_tracked), _p(stats[nws_off:]),
                                    _p(ps), N, C, HW, float(bn.eps), float(bn.momentum), act, BF16, _stream()),
            "cot_bn_act_forward_ps")
        return
    _ck(L.cot_bn_act_forward(_p(x), _p(residual), _p(y), _p(bn.weight), _p(bn.bias), _p(stats), _p(stats[C:]),
                             _p(bn.running_mean), _p(bn.running_var), _p(bn.num_batches_tracked), _p(stats[nws_off:]),
                             N, C, HW, float(bn.eps), float(bn.momentum), act, BF16, _stream()), "cot_bn_act_forward")


# BatchNorm statistics out of the producing 1x1 convolution's epilogue (SURVEY 7.6 / DESIGN 4.9c; opt-in COT_BN_EPILOGUE=1): on
# planes of more than 256 pixels the convolution also writes per-tile sums of what it stores, a one-wave-per-channel finalize turns
# them into mean / rstd (+ running statistics), and the flat apply kernel normalises -- the statistics pass over the tensor is gone
BN_EPILOGUE = os.environ.get("COT_BN_EPILOGUE", "0") == "1"  # measured +0.63 ms per CoTNet-50 step (profiles/r05_bn_epilogue_stats_ab.log): off
_EPI_OK = _lib.register_cache({})


def _epi_ok(L, Ci, c1, two, HW):
    k = (Ci, c1, two, HW)
    v = _EPI_OK.get(k)
    if v is None:
        v = _EPI_OK[k] = bool(L.cot_conv1x1_stats_covers(Ci, c1, 1 if two else 0, HW))
    return v


def _conv_bn_fwd(L, x1, x2, c1, conv, y_pre, y, bn, stats, nws_off, N, Ci, Co, HW, act, residual=None, ps=None, mask=None):
    """y_pre = conv1x1([x1 | x2]); y = act(bn(y_pre) [+ residual]) -- the models' conv -> BatchNorm pairs (models/cotnet.py:51-62,
    :228-264).  stats: fp32 [2*Co + workspace] as for _bn_fwd."""
    st = _stream()
    bias = conv.bias
    if BN_EPILOGUE and ps is None and _epi_ok(L, Ci, c1, x2 is not None, HW):
        part = torch.empty(int(L.cot_gn9_stats_floats(N, Co, HW)), dtype=torch.float32, device=y.device)
        _ck(L.cot_conv1x1_forward_stats(_p(x1), _p(x2), c1, _p(conv.weight), _p(bias), _p(y_pre), _p(part), N, Ci, Co, HW, BF16, st),
            "cot_conv1x1_forward_stats")
        _ck(L.cot_bn_tile_stats_finalize(_p(part), _p(stats), _p(stats[Co:]), _p(bn.running_mean), _p(bn.running_var),
                                         _p(bn.num_batches_tracked), N, Co, HW, float(bn.eps), float(bn.momentum), st),
            "cot_bn_tile_stats_finalize")
        _ck(L.cot_bn_act_apply_forward(_p(y_pre), _p(residual), _p(y), _p(mask), _p(bn.weight), _p(bn.bias), _p(stats), _p(stats[Co:]),
                                       N, Co, HW, act, BF16, st), "cot_bn_act_apply_forward")
        return
    _ck(L.cot_conv1x1_forward(_p(x1), _p(x2), c1, _p(conv.weight), _p(bias), _p(y_pre), N, Ci, Co, HW, BF16, st), "cot_conv1x1_forward")
    _bn_fwd(L, y_pre, y, bn, stats, nws_off, N, Co, HW, act, residual=residual, ps=ps, mask=mask)


def _bn_bwd(L, dy, x, y, dx, bn, stats, N, C, HW, act, nws, dres=None, ps=None, mask=None):
    """-> (dgamma, dbeta): the parameters' slots in the flat gradient buckets when registered (grad_sink), else fresh.
    mask: the sign mask the forward wrote (then `y` is not read)"""
    dg, db = grad_sink.out_like(bn.weight), grad_sink.out_like(bn.bias)
    ws = torch.empty(max(nws, 1), dtype=torch.float32, device=dy.device)
    if mask is not None:
        _ck(L.cot_bn_act_backward_mask(_p(dy), _p(x), _p(mask), _p(dx), _p(dres), _p(bn.weight), _p(bn.bias), _p(stats),
                                       _p(stats[C:]), _p(dg), _p(db), _p(ws), _p(ps), N, C, HW, act, BF16, _stream()),
            "cot_bn_act_backward_mask")
        return dg, db
    if ps is not None:
        _ck(L.cot_bn_act_backward_ps(_p(dy), _p(x), _p(y), _p(dx), _p(dres), _p(bn.weight), _p(bn.bias), _p(stats),
                                     _p(stats[C:]), _p(dg), _p(db), _p(ws), _p(ps), N, C, HW, act, BF16, _stream()),
            "cot_bn_act_backward_ps")
        return dg, db
    _ck(L.cot_bn_act_backward(_p(dy), _p(x), _p(y), _p(dx), _p(dres), _p(bn.weight), _p(bn.bias), _p(stats), _p(stats[C:]),
                              _p(dg), _p(db), _p(ws), N, C, HW, act, BF16, _stream()),
        "cot_bn_act_backward")
    return dg, db


# ---- stochastic depth (models/cotnet.py:256-257; models/layers/drop.py:140-168: per sample, the block's branch is dropped with
# probability p and scaled by 1 / (1 - p) otherwise) inside the single-node Bottleneck: the per-sample scale goes into the
# bn3 + residual + ReLU kernels (cot_bn_act_*_ps).  One vectorised draw per step serves every block of a model
# (prepare_drop_path, called by ResNet.forward_features): 3 small launches per step instead of 4 per block.
_DP_PLANS = weakref.WeakKeyDictionary()  # model -> (blocks with an active DropPath, their keep probabilities on the device)


def prepare_drop_path(model, x):
    if not (ENABLED and model.training):
        return
    ent = _DP_PLANS.get(model)
    if ent is None or ent[1].device != x.device:
        blocks = [m for m in model.modules() if getattr(m, "drop_path", None) is not None and getattr(m.drop_path, "drop_prob", 0.0)]
        keep = torch.tensor([1.0 - float(b.drop_path.drop_prob) for b in blocks], dtype=torch.float32, device=x.device)
        ent = _DP_PLANS[model] = (blocks, keep)
    blocks, keep = ent
    if not blocks:
        return
    u = torch.rand((len(blocks), x.shape[0]), dtype=torch.float32, device=x.device)
    ps_all = u.add_(keep[:, None]).floor_().div_(keep[:, None])  # 0 or 1 / keep  (drop.py:165-167)
    for i, b in enumerate(blocks):
        b._ps_next = ps_all[i]


def _drop_path_scale(blk, N, dev):
    """this forward's per-sample scale for `blk` (None: no stochastic depth); consumes what prepare_drop_path laid out"""
    dp = blk.drop_path
    if dp is None or not dp.drop_prob or not blk.training:
        return None
    ps = getattr(dp, "fixed_scale", None)  # (tests: a module that applies a GIVEN per-sample scale)
    if ps is not None:
        return ps.to(device=dev, dtype=torch.float32).contiguous()
    ps = getattr(blk, "_ps_next", None)
    blk._ps_next = None
    if ps is None or ps.shape[0] != N or ps.device != dev:  # (block used outside a ResNet.forward: draw its own)
        keep = 1.0 - float(dp.drop_prob)
        ps = torch.rand(N, dtype=torch.float32, device=dev).add_(keep).floor_().div_(keep)
    return ps


def _cot_forward(L, layer, x):
    """the layer's forward on the library kernels -> (out, tensors to keep for backward, aggregation geometry)"""
    N, C, H, W = x.shape
    HW, Ch, Ce = H * W, C // 2, 9 * C // 8
    dev = x.device
    pl = _plan(layer)
    A = pl.se0.out_channels
    GX = pl.grouped
    ws_bytes, nws_c, nws_h, nws_a = _sizes(L, N, C, H, W, A, pl.ke0.groups, GX)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    masks = _masks(L, H, W, dev)
    st = _stream()
    new = lambda c: torch.empty((N, c, H, W), dtype=x.dtype, device=dev)  # noqa: E731
    stat = lambda c, nws: torch.empty(2 * c + nws, dtype=torch.float32, device=dev)  # noqa: E731

    # static context k = relu(bn(conv3x3_grouped(x)))                                             (ref :80)
    k_pre, k = new(C), new(C)
    _conv3x3_fwd(L, pl.ke0, x, k_pre, masks, ws, N, C, pl.ke0.groups, H, W)
    s_k = stat(C, nws_c)
    _bn_fwd(L, k_pre, k, pl.ke1, s_k, 2 * C, N, C, HW, 1)
    # attention logits from [x | k]                                                             (ref :81-85)
    e0, e1 = new(Ch), new(Ch)
    qk = None
    if GX:  # CoXtLayer: [x0, k0, x1, k1, ...] so that each of the two groups sees matching halves of x and k (ref :153-154)
        qk = torch.stack([x, k], dim=2).view(N, 2 * C, H, W)
        _ck(L.cot_conv1x1g_forward(_p(qk), _p(pl.em0.weight), None, _p(e0), N, 2 * C, Ch, 2, HW, BF16, st), "cot_conv1x1g_forward")
    s_e = stat(Ch, nws_h)
    if GX:
        _bn_fwd(L, e0, e1, pl.em1, s_e, 2 * Ch, N, Ch, HW, 1)
    else:
        _conv_bn_fwd(L, x, k, C, pl.em0, e0, e1, pl.em1, s_e, 2 * Ch, N, 2 * C, Ch, HW, 1)
    e3 = new(Ce)
    gn = pl.gn
    # GroupNorm of the logits fused into its neighbours: statistics out of embed[3]'s epilogue, normalisation in the
    # aggregation's prologue -- the normalised tensor `w` is never materialised (stages with planes of more than 256 pixels)
    fused_gn = GN_FUSED and not GX and _gn_fused_ok(L, Ch, HW, W)
    if GX:
        _ck(L.cot_conv1x1g_forward(_p(e1), _p(pl.em3.weight), _p(pl.em3.bias), _p(e3), N, Ch, Ce, 2, HW, BF16, st),
            "cot_conv1x1g_forward")
    elif fused_gn:
        part = torch.empty(int(L.cot_gn9_stats_floats(N, Ce, HW)), dtype=torch.float32, device=dev)
        _ck(L.cot_conv1x1_forward_gn9(_p(e1), None, Ch, _p(pl.em3.weight), _p(pl.em3.bias), _p(e3), _p(part), N, Ch, Ce, HW, BF16, st),
            "cot_conv1x1_forward_gn9")
    else:
        _ck(L.cot_conv1x1_forward(_p(e1), None, Ch, _p(pl.em3.weight), _p(pl.em3.bias), _p(e3), N, Ch, Ce, HW, BF16, st),
            "cot_conv1x1_forward")
    if fused_gn:
        w = None
        gn_mean = torch.empty(2 * N * gn.num_groups, dtype=torch.float32, device=dev)
        gn_rstd = gn_mean[N * gn.num_groups:]
        _ck(L.cot_gn9_stats_finalize(_p(part), _p(gn_mean), _p(gn_rstd), N, Ce, HW, float(gn.eps), st), "cot_gn9_stats_finalize")
    elif HW <= 8192:  # one (image, group) fits a workgroup's registers: 1 read + 1 write (csrc/group_norm9.hip)
        w = new(Ce)
        gn_mean = torch.empty(2 * N * gn.num_groups, dtype=torch.float32, device=dev)
        gn_rstd = gn_mean[N * gn.num_groups:]
        _ck(L.cot_group_norm9_forward(_p(e3), _p(gn.weight), _p(gn.bias), _p(w), _p(gn_mean), _p(gn_rstd), N, Ce, HW,
                                      float(gn.eps), BF16, st), "cot_group_norm9_forward")
    else:
        w, gn_mean, gn_rstd = torch.native_group_norm(e3, gn.weight, gn.bias, N, Ce, HW, gn.num_groups, gn.eps)
    # values                                                                                     (ref :87)
    v_pre, v = new(C), new(C)
    if GX:
        _ck(L.cot_conv1x1g_forward(_p(x), _p(pl.cv0.weight), None, _p(v_pre), N, C, C, 2, HW, BF16, st), "cot_conv1x1g_forward")
    s_v = stat(C, nws_c)
    if GX:
        _bn_fwd(L, v_pre, v, pl.cv1, s_v, 2 * C, N, C, HW, 0)
    else:
        _conv_bn_fwd(L, x, None, C, pl.cv0, v_pre, v, pl.cv1, s_v, 2 * C, N, C, C, HW, 0)
    # local aggregation, bn + swish                                                              (ref :88-90)
    # (CoXtLayer folds its two groups into the batch: [N, C] -> [2N, C/2], weights [2N, 1, C/16, 9]: views of the same memory)
    geom = _lib.AggGeom(2 * N, C // 2, H, W, 1, C // 16, 3, 3, 1, 1, 1, 1, 1, 1) if GX else \
        _lib.AggGeom(N, C, H, W, 1, C // 8, 3, 3, 1, 1, 1, 1, 1, 1)
    a, y = new(C), (None if BN_TAIL else new(C))
    s_y = stat(C, nws_c)
    bnl = pl.bn
    if BN_TAIL:  # (aggregation + the statistics of bn; bn + swish themselves happen inside the tail's kernels)
        y_final = _agg_fwd_stats(L, v, e3 if fused_gn else w, a, gn if fused_gn else None, gn_mean, gn_rstd, geom, bnl, s_y, N, C, H, W)
    else:
        if fused_gn:
            _ck(L.cot_agg_gn9_forward(_p(v), _p(e3), _p(gn_mean), _p(gn_rstd), _p(gn.weight), _p(gn.bias), gn.num_groups, _p(a),
                                      ctypes.byref(geom), BF16, st), "cot_agg_gn9_forward")
        else:
            _ck(L.cot_agg_forward(_p(v), _p(w), _p(a), ctypes.byref(geom), BF16, _lib.COT_NCHW, st), "cot_agg_forward")
        _bn_fwd(L, a, y, bnl, s_y, 2 * C, N, C, HW, 2)
    # radix-2 split attention                                                                    (ref :92-104)
    # descriptors are kept channel-major ([C][N]) so that the se branch runs on the 1x1-convolution / BatchNorm
    # kernels with the batch as the pixel axis
    row = lambda c: torch.empty((c, N), dtype=x.dtype, device=dev)  # noqa: E731
    gapT, hpre, h, logitsT = row(C), row(A), row(A), row(2 * C)
    if BN_TAIL:
        _tail_gap(L, a, k, gapT, bnl, s_y, y_final, N, C, HW, 0)
    else:
        _ck(L.cot_radix_gap_t(_p(y), _p(k), _p(gapT), N, C, HW, BF16, st), "cot_radix_gap_t")
    _ck(L.cot_conv1x1_forward(_p(gapT), None, C, _p(pl.se0.weight), _p(pl.se0.bias), _p(hpre), 1, C, A, N, BF16, st),
        "cot_conv1x1_forward")
    s_a = stat(A, nws_a)
    _bn_fwd(L, hpre, h, pl.sebn, s_a, 2 * A, 1, A, N, 1)
    _ck(L.cot_conv1x1_forward(_p(h), None, A, _p(pl.se3.weight), _p(pl.se3.bias), _p(logitsT), 1, A, 2 * C, N, BF16,
                              st), "cot_conv1x1_forward")
    attn = torch.empty((N, C, 2), dtype=x.dtype, device=dev)
    out = new(C)
    if BN_TAIL:
        _ck(L.cot_radix_mix_logits_bn(_p(a), _p(k), _p(logitsT), _p(out), _p(attn), _p(bnl.weight), _p(bnl.bias), _p(s_y), _p(s_y[C:]),
                                      N, C, HW, 0, BF16, st), "cot_radix_mix_logits_bn")
    else:
        _ck(L.cot_radix_mix_logits(_p(y), _p(k), _p(logitsT), _p(out), _p(attn), N, C, HW, BF16, st),
            "cot_radix_mix_logits")

    return out, (x, k_pre, k, e0, e1, e3, w, gn_mean, gn_rstd, v_pre, v, a, y, attn, s_k, s_e, s_v, s_y, gapT, hpre, h,
                 s_a, qk), geom


_N_SAVED = 23  # tensors _cot_forward hands back for the backward pass (the last one, qk, is None for a CotLayer)


def _cot_backward(L, layer, saved, geom, gout, side=None):
    """-> (dx, gradients of _Plan.params in that order).  `side`: the caller's side stream for the weight gradients (a
    Bottleneck node passes its own and joins it itself); None = this call opens and joins one."""
    (x, k_pre, k, e0, e1, e3, w, gn_mean, gn_rstd, v_pre, v, a, y, attn, s_k, s_e, s_v, s_y, gapT, hpre, h,
     s_a, qk) = saved
    N, C, H, W = x.shape
    HW, Ch, Ce = H * W, C // 2, 9 * C // 8
    dev = x.device
    pl = _plan(layer)
    A = pl.se0.out_channels
    GX = pl.grouped
    ws_bytes, nws_c, nws_h, nws_a = _sizes(L, N, C, H, W, A, pl.ke0.groups, GX)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    masks = _masks(L, H, W, dev)
    st = _stream()
    own_side = side is None
    if own_side:
        side = _Side(dev, ws_bytes, ws, pl.params)
    ke0, ke1, em0, em1, em3, cv0, cv1 = pl.ke0, pl.ke1, pl.em0, pl.em1, pl.em3, pl.cv0, pl.cv1
    se0, sebn, se3 = pl.se0, pl.sebn, pl.se3
    gout = gout.contiguous()

    # radix mix -> pair-softmax backward -> se branch (two 1x1 convolutions over the batch axis) -> gap
    row = lambda c: torch.empty((c, N), dtype=x.dtype, device=dev)  # noqa: E731
    glogT, gh, ggapT = row(2 * C), row(A), row(C)
    bnl = pl.bn
    if y is None:  # (the forward folded bn + swish into the tail: so does the backward)
        tsum = torch.empty(N * C * 4, dtype=torch.float32, device=dev)
        _ck(L.cot_radix_mix_backward_reduce_bn(_p(gout), _p(a), _p(k), _p(attn), _p(glogT), _p(tsum), _p(bnl.weight), _p(bnl.bias),
                                               _p(s_y), _p(s_y[C:]), N, C, HW, 0, BF16, st), "cot_radix_mix_backward_reduce_bn")
    else:
        _ck(L.cot_radix_mix_backward_reduce(_p(gout), _p(y), _p(k), _p(attn), _p(glogT), N, C, HW, BF16, st),
            "cot_radix_mix_backward_reduce")
    _ck(L.cot_conv1x1_backward_data(_p(glogT), _p(se3.weight), _p(gh), None, A, 0, _p(ws), 1, A, 2 * C, N, BF16, st),
        "cot_conv1x1_backward_data")
    g_w3, g_b3 = grad_sink.out_like(se3.weight), grad_sink.out_like(se3.bias)
    side.run(lambda st_, a_=(_p(glogT), _p(h), None, A, _p(g_w3), _p(g_b3), _p(side.ws), 1, A, 2 * C, N, BF16): _ck(L.cot_conv1x1_backward_weight(*a_, st_), "cot_conv1x1_backward_weight"), glogT, h)
    ghpre = row(A)
    d_sa_w, d_sa_b = _bn_bwd(L, gh, hpre, None, ghpre, sebn, s_a, 1, A, N, 1, nws_a)
    _ck(L.cot_conv1x1_backward_data(_p(ghpre), _p(se0.weight), _p(ggapT), None, C, 0, _p(ws), 1, C, A, N, BF16, st),
        "cot_conv1x1_backward_data")
    g_w0, g_b0 = grad_sink.out_like(se0.weight), grad_sink.out_like(se0.bias)
    side.run(lambda st_, a_=(_p(ghpre), _p(gapT), None, C, _p(g_w0), _p(g_b0), _p(side.ws), 1, C, A, N, BF16): _ck(L.cot_conv1x1_backward_weight(*a_, st_), "cot_conv1x1_backward_weight"), ghpre, gapT)
    # bn + swish, aggregation
    ga, gk = torch.empty_like(a), torch.empty_like(k)
    if y is None:
        d_bn_w, d_bn_b = grad_sink.out_like(bnl.weight), grad_sink.out_like(bnl.bias)
        _ck(L.cot_radix_mix_backward_apply_bn(_p(gout), _p(a), _p(attn), _p(ggapT), _p(tsum), _p(ga), _p(gk), _p(bnl.weight), _p(bnl.bias),
                                              _p(s_y), _p(s_y[C:]), _p(d_bn_w), _p(d_bn_b), N, C, HW, 0, BF16, st),
            "cot_radix_mix_backward_apply_bn")
    else:
        gy = torch.empty_like(y)
        _ck(L.cot_radix_mix_backward_apply(_p(gout), _p(attn), _p(ggapT), _p(gy), _p(gk), N, C, HW, BF16, st),
            "cot_radix_mix_backward_apply")
        d_bn_w, d_bn_b = _bn_bwd(L, gy, a, None, ga, bnl, s_y, N, C, HW, 2, nws_c)
    gv, gw = torch.empty_like(v), torch.empty_like(e3)
    if w is None:  # (the forward normalised the logits inside the aggregation: so does the backward; gw = d / d normalised weights)
        gn_ = pl.gn
        _ck(L.cot_agg_gn9_backward(_p(ga), _p(v), _p(e3), _p(gn_mean), _p(gn_rstd), _p(gn_.weight), _p(gn_.bias), gn_.num_groups,
                                   _p(gv), _p(gw), ctypes.byref(geom), BF16, st), "cot_agg_gn9_backward")
    else:
        _ck(L.cot_agg_backward(_p(ga), _p(v), _p(w), _p(gv), _p(gw), ctypes.byref(geom), BF16, _lib.COT_NCHW, st),
            "cot_agg_backward")
    # values branch: bn, conv1x1 -> first contribution to dx
    gv_pre = ga  # (reuse: ga is dead)
    d_cv_w, d_cv_b = _bn_bwd(L, gv, v_pre, None, gv_pre, cv1, s_v, N, C, HW, 0, nws_c)
    gx = torch.empty_like(x)
    g_wv = grad_sink.out_like(cv0.weight)
    if GX:
        _ck(L.cot_conv1x1g_backward_data(_p(gv_pre), _p(cv0.weight), _p(gx), 0, N, C, C, 2, HW, BF16, st), "cot_conv1x1g_backward_data")
        side.run(lambda st_, a_=(_p(gv_pre), _p(x), _p(g_wv), None, _p(side.ws), N, C, C, 2, HW, BF16): _ck(L.cot_conv1x1g_backward_weight(*a_, st_), "cot_conv1x1g_backward_weight"), gv_pre, x)
    else:
        _ck(L.cot_conv1x1_backward_data(_p(gv_pre), _p(cv0.weight), _p(gx), None, C, 0, _p(ws), N, C, C, HW, BF16, st),
            "cot_conv1x1_backward_data")
        side.run(lambda st_, a_=(_p(gv_pre), _p(x), None, C, _p(g_wv), None, _p(side.ws), N, C, C, HW, BF16): _ck(L.cot_conv1x1_backward_weight(*a_, st_), "cot_conv1x1_backward_weight"), gv_pre, x)
    # logits branch: GroupNorm, conv1x1(+bias), bn+relu, conv1x1 on [x | k] -> dx +=, dk +=
    gn = pl.gn
    if HW <= 8192:
        ge3, g_gn_w, g_gn_b = torch.empty_like(e3), grad_sink.out_like(gn.weight), grad_sink.out_like(gn.bias)
        gn_ws = torch.empty(2 * N * Ce, dtype=torch.float32, device=dev)
        # (dx here; dgamma / dbeta -- a launch of their own -- beside the weight gradients)
        _ck(L.cot_group_norm9_backward(_p(gw), _p(e3), _p(gn_mean), _p(gn_rstd), _p(gn.weight), _p(ge3), None, None, _p(gn_ws), N, Ce, HW,
                                       BF16, st), "cot_group_norm9_backward")
        side.run(lambda st_, a_=(_p(gn_ws), _p(g_gn_w), _p(g_gn_b), N, Ce, BF16): _ck(L.cot_group_norm9_backward_params(*a_, st_), "cot_group_norm9_backward_params"), gn_ws)
    else:
        ge3, g_gn_w, g_gn_b = torch.ops.aten.native_group_norm_backward(
            gw, e3, gn_mean, gn_rstd, gn.weight, N, Ce, HW, gn.num_groups, [True, True, True])
        ge3 = ge3.contiguous()
    ge1 = torch.empty_like(e1)
    g_we3, g_be3 = grad_sink.out_like(em3.weight), grad_sink.out_like(em3.bias)
    if GX:
        _ck(L.cot_conv1x1g_backward_data(_p(ge3), _p(em3.weight), _p(ge1), 0, N, Ch, Ce, 2, HW, BF16, st), "cot_conv1x1g_backward_data")
        side.run(lambda st_, a_=(_p(ge3), _p(e1), _p(g_we3), _p(g_be3), _p(side.ws), N, Ch, Ce, 2, HW, BF16): _ck(L.cot_conv1x1g_backward_weight(*a_, st_), "cot_conv1x1g_backward_weight"), ge3, e1)
    else:
        _ck(L.cot_conv1x1_backward_data(_p(ge3), _p(em3.weight), _p(ge1), None, Ch, 0, _p(ws), N, Ch, Ce, HW, BF16,
                                        st), "cot_conv1x1_backward_data")
        side.run(lambda st_, a_=(_p(ge3), _p(e1), None, Ch, _p(g_we3), _p(g_be3), _p(side.ws), N, Ch, Ce, HW, BF16): _ck(L.cot_conv1x1_backward_weight(*a_, st_), "cot_conv1x1_backward_weight"), ge3, e1)
    ge0 = torch.empty_like(e0)
    d_em_w, d_em_b = _bn_bwd(L, ge1, e0, None, ge0, em1, s_e, N, Ch, HW, 1, nws_h)  # (ReLU mask recomputed from e0)
    g_we0 = grad_sink.out_like(em0.weight)
    if GX:  # gradient of the interleaved [x0, k0, x1, k1, ...]: de-interleaved into dx / dk (two strided adds)
        gqk = torch.empty_like(qk)
        _ck(L.cot_conv1x1g_backward_data(_p(ge0), _p(em0.weight), _p(gqk), 0, N, 2 * C, Ch, 2, HW, BF16, st), "cot_conv1x1g_backward_data")
        gq5 = gqk.view(N, C, 2, H, W)
        gx.add_(gq5[:, :, 0])
        gk.add_(gq5[:, :, 1])
        side.run(lambda st_, a_=(_p(ge0), _p(qk), _p(g_we0), None, _p(side.ws), N, 2 * C, Ch, 2, HW, BF16): _ck(L.cot_conv1x1g_backward_weight(*a_, st_), "cot_conv1x1g_backward_weight"), ge0, qk)
    else:
        _ck(L.cot_conv1x1_backward_data(_p(ge0), _p(em0.weight), _p(gx), _p(gk), C, 3, _p(ws), N, 2 * C, Ch, HW, BF16,
                                        st), "cot_conv1x1_backward_data")
        side.run(lambda st_, a_=(_p(ge0), _p(x), _p(k), C, _p(g_we0), None, _p(side.ws), N, 2 * C, Ch, HW, BF16): _ck(L.cot_conv1x1_backward_weight(*a_, st_), "cot_conv1x1_backward_weight"), ge0, x, k)
    # key branch: bn+relu, grouped 3x3 -> dx +=
    gk_pre = gv  # (reuse: gv is dead)
    d_ke_w, d_ke_b = _bn_bwd(L, gk, k_pre, None, gk_pre, ke1, s_k, N, C, HW, 1, nws_c)
    G = ke0.groups
    g_wk = grad_sink.out_like(ke0.weight)
    side.run(lambda st_, a_=(_p(gk_pre), _p(x), _p(g_wk), _p(masks), _p(side.ws), N, C, C, G, H, W, BF16,
                                               _guard_elems(x)): _ck(L.cot_conv3x3g_backward_weight_guarded(*a_, st_), "cot_conv3x3g_backward_weight"), gk_pre, x, masks)
    _conv3x3_dgrad(L, ke0, gk_pre, gx, 1, masks, ws, N, C, G, H, W)
    if own_side:
        side.join()
    # order = _Plan.params
    return gx, (g_wk, d_ke_w, d_ke_b, g_we0, d_em_w, d_em_b, g_we3, g_be3, g_gn_w, g_gn_b, g_wv, d_cv_w, d_cv_b,
                d_bn_w, d_bn_b, g_w0, g_b0, d_sa_w, d_sa_b, g_w3, g_b3)


class _CotLayerNode(Function):
    @staticmethod
    @_one_stream_query
    def forward(ctx, layer, x, *params):
        # params (_Plan.params) are only here so that autograd routes their gradients; values are read off `layer`
        out, saved, geom = _cot_forward(_lib.lib(), layer, x)
        ctx.layer, ctx.geom = layer, geom
        ctx.save_for_backward(*saved)
        return out

    @staticmethod
    @_one_stream_query
    def backward(ctx, gout):
        gx, gparams = _cot_backward(_lib.lib(), ctx.layer, ctx.saved_tensors, ctx.geom, gout)
        return (None, gx) + gparams


def _bn_static_ok(bn):
    return (isinstance(bn, nn.BatchNorm2d) and bn.affine and bn.track_running_stats and bn.momentum is not None
            and bn.num_batches_tracked is not None)


def _conv_ok(conv, k, groups=None):
    return (isinstance(conv, nn.Conv2d) and conv.kernel_size == (k, k) and conv.stride == (1, 1)
            and conv.padding == (k // 2, k // 2) and conv.dilation == (1, 1) and (groups is None or conv.groups == groups))


def eligible(layer, x):
    """training-mode CotLayer on a bf16 NCHW tensor whose every piece the kernels cover (mixed precision as
    cotnet_amd.flat_sgd.to_mixed_bf16 sets it up: bf16 convolution / GroupNorm parameters, fp32 BatchNorm parameters)"""
    if not (ENABLED and layer.training and (x.is_cuda or not _DEVICE_ONLY) and x.dim() == 4
            and x.dtype == torch.bfloat16 and x.is_contiguous() and x.data_ptr() % 16 == 0
            and x.shape[1] == layer.dim):
        return False
    pl = _plan(layer)
    return (pl.static_ok and pl.ke0.weight.dtype == torch.bfloat16 and pl.em3.weight.dtype == torch.bfloat16
            and pl.gn.weight.dtype == torch.bfloat16 and pl.bn.weight.dtype == torch.float32 and pl.bn.training
            and pl.ke1.training)


# how many times each node kind ran since the last reset (bench.py reports what actually executed, per step)
NODE_COUNTS = {"cot_layer": 0, "bottleneck": 0, "bottleneck_channel_major": 0, "split_attn_block": 0, "bottleneck_eval": 0}


def reset_node_counts():
    for k in NODE_COUNTS:
        NODE_COUNTS[k] = 0


def cot_layer_forward(layer, x):
    """layer(x) through the single-node path; caller checks `eligible` first"""
    NODE_COUNTS["cot_layer"] += 1
    return _CotLayerNode.apply(layer, x, *_plan(layer).params)


# ---- the whole Bottleneck (models/cotnet.py:228-264; models/cotnet_hybrid.py:172-202 for SE-CoTNetD's CoT blocks, the same
# sequence) as one node: conv1 -> bn1+relu [-> 3x3/2 average pooling "avd"] ->
# CotLayer -> conv3 -> bn3 + residual + relu.  The residual branch may carry the stage's 1x1 projection (`downsample` =
# [Identity,] conv1x1, BatchNorm); in a stride-2 block that projection has stride 2 = the stride-1 convolution on every
# second pixel.  dx collects the residual gradient, the projection's and conv1's data gradients inside the kernels
# (`accumulate`).
class _BlockPlan:
    __slots__ = ("conv1", "bn1", "cot", "conv3", "bn3", "ds_conv", "ds_bn", "ds_stride", "avd", "avd_post", "ds_pool2", "params",
                 "static_ok")

    def __init__(self, blk):
        from .cotnet import CotLayer, CoXtLayer
        self.conv1, self.bn1, self.cot, self.conv3, self.bn3 = blk.conv1, blk.bn1, blk.conv2, blk.conv3, blk.bn3
        ds = blk.downsample
        self.ds_conv = self.ds_bn = None
        self.ds_stride = 1
        from .layers import BlurPool2d
        avd = blk.avd
        # cotnet_hybrid.CoTBottleneck pools AFTER the layer when avd_first is False (models/cotnet_hybrid.py:196-199): SE-CoTNetD-152's
        # BlurPool2d(filt_size 3, stride 2) behind conv2 -- `avd_post`; cotnet.Bottleneck pools in front of it (3 x 3 / 2 average) -- `avd`
        post = (avd is not None and not getattr(blk, "avd_first", True) and isinstance(avd, BlurPool2d) and avd.filt_size == 3
                and avd.stride == 2)
        self.avd_post = post
        self.avd = avd is not None and not post
        avd_ok = avd is None or post or (isinstance(avd, nn.AvgPool2d) and avd.kernel_size == 3 and avd.stride == 2
                                         and avd.padding == 1 and not avd.ceil_mode and avd.count_include_pad
                                         and avd.divisor_override is None and getattr(blk, "avd_first", True))
        ds_ok = ds is None and avd is None
        self.ds_pool2 = False
        if isinstance(ds, nn.Sequential) and (len(ds) == 2 or (len(ds) == 3 and isinstance(ds[0], (nn.Identity, nn.AvgPool2d)))):
            self.ds_conv, self.ds_bn = ds[-2], ds[-1]  # models/resnet.py:364-394: [pool,] conv, norm
            c = ds[-2]
            pool = ds[0] if len(ds) == 3 and isinstance(ds[0], nn.AvgPool2d) else None
            # `avg_down` shortcut of a stride-2 block (models/resnet.py:380-394): AvgPool2d(2, 2) in front of a stride-1 projection
            self.ds_pool2 = (pool is not None and pool.kernel_size == 2 and pool.stride == 2 and pool.padding == 0
                             and pool.divisor_override is None)  # (ceil_mode / count_include_pad: no effect on even planes, checked at run time)
            self.ds_stride = 2 if (avd is not None and pool is None) else 1
            ds_ok = (isinstance(c, nn.Conv2d) and c.kernel_size == (1, 1) and c.padding == (0, 0) and c.groups == 1
                     and c.stride == (self.ds_stride, self.ds_stride) and c.bias is None and _bn_static_ok(ds[-1])
                     and (pool is None or (self.ds_pool2 and avd is not None)))
        self.static_ok = (
            ds_ok and avd_ok and isinstance(blk.conv2, (CotLayer, CoXtLayer)) and blk.drop_block is None
            and (blk.drop_path is None or hasattr(blk.drop_path, "drop_prob")) and getattr(blk, "se", None) is None
            and isinstance(blk.act1, nn.ReLU)
            and isinstance(blk.act3, nn.ReLU) and _conv_ok(blk.conv1, 1, 1) and blk.conv1.bias is None
            and _conv_ok(blk.conv3, 1, 1) and blk.conv3.bias is None and _bn_static_ok(blk.bn1)
            and _bn_static_ok(blk.bn3) and blk.conv1.in_channels % 8 == 0 and blk.conv3.out_channels % 8 == 0
            and _plan(blk.conv2).static_ok)
        self.params = [blk.conv1.weight, blk.bn1.weight, blk.bn1.bias] + _plan(blk.conv2).params + \
            [blk.conv3.weight, blk.bn3.weight, blk.bn3.bias] + \
            ([self.ds_conv.weight, self.ds_bn.weight, self.ds_bn.bias] if self.ds_conv is not None else [])


def _block_plan(blk):
    p = _BLOCK_PLANS.get(blk)
    if p is None:
        p = _BLOCK_PLANS[blk] = _BlockPlan(blk)
    return p


_BSIZES = _lib.register_cache({})


def _block_sizes(L, N, Cin, Cw, Cout, HW):
    k = (N, Cin, Cw, Cout, HW)
    v = _BSIZES.get(k)
    if v is None:
        ws = max(int(L.cot_conv1x1_workspace(N, Cin, Cw, HW, 0)), int(L.cot_conv1x1_workspace(N, Cw, Cout, HW, 0)),
                 int(L.cot_conv1x1_workspace(N, Cin, Cout, HW, 0)))
        v = _BSIZES[k] = (ws, int(L.cot_bn_act_workspace(N, Cw)), int(L.cot_bn_act_workspace(N, Cout)))
    return v


class _BottleneckNode(Function):
    @staticmethod
    @_one_stream_query
    def forward(ctx, blk, x, *params):
        L = _lib.lib()
        bp = _block_plan(blk)
        N, Cin, H, W = x.shape
        Cw, Cout = bp.conv1.out_channels, bp.conv3.out_channels
        Ho, Wo = ((H - 1) // 2 + 1, (W - 1) // 2 + 1) if (bp.avd or bp.avd_post) else (H, W)  # 3x3/2 pooling, padding 1 / BlurPool
        HW, HWo = H * W, Ho * Wo
        dev, st = x.device, _stream()
        _, nws_w, _ = _block_sizes(L, N, Cin, Cw, Cout, HW)
        _, _, nws_o = _block_sizes(L, N, Cin, Cw, Cout, HWo)
        new = lambda c, h, w: torch.empty((N, c, h, w), dtype=x.dtype, device=dev)  # noqa: E731
        stat = lambda c, nws: torch.empty(2 * c + nws, dtype=torch.float32, device=dev)  # noqa: E731
        # (the CoT layer's input -- a1, or its pooled version -- is what the grouped 3x3 weight gradient reads shifted: margins)
        c1, a1 = new(Cw, H, W), (new(Cw, H, W) if bp.avd else _new_guarded(N, Cw, H, W, x.dtype, dev))
        s_1 = stat(Cw, nws_w)
        _conv_bn_fwd(L, x, None, Cin, bp.conv1, c1, a1, bp.bn1, s_1, 2 * Cw, N, Cin, Cw, HW, 1)
        if bp.avd:
            p1 = _new_guarded(N, Cw, Ho, Wo, x.dtype, dev)
            _ck(L.cot_avgpool3x3s2_forward(_p(a1), _p(p1), N * Cw, H, W, BF16, st), "cot_avgpool3x3s2_forward")
        else:
            p1 = a1
        cot_out, saved, geom = _cot_forward(L, bp.cot, p1)
        if bp.avd_post:  # anti-aliased down-sampling behind the layer (cotnet_hybrid.py:196-199; blur_pool.py:53-58)
            cot_full = cot_out
            cot_out = new(Cw, Ho, Wo)
            _ck(L.cot_blurpool3x3s2_forward(_p(cot_full), _p(cot_out), N * Cw, H, W, BF16, st), "cot_blurpool3x3s2_forward")
        c3, y = new(Cout, Ho, Wo), new(Cout, Ho, Wo)
        if bp.ds_conv is not None:  # projection shortcut: bn(conv1x1(x)), on every second pixel in a stride-2 block
            if bp.ds_pool2:  # `avg_down`: 2 x 2 average pooling, then the stride-1 projection
                xs = torch.empty((N, Cin, H // 2, W // 2), dtype=x.dtype, device=dev)
                _ck(L.cot_avgpool2x2s2_forward(_p(x), _p(xs), N * Cin, H, W, BF16, st), "cot_avgpool2x2s2_forward")
            elif bp.ds_stride == 2 and H % 2 == 0 and W % 2 == 0:  # every second pixel: one pass, 16-byte accesses (pool3x3.hip)
                xs = torch.empty((N, Cin, H // 2, W // 2), dtype=x.dtype, device=dev)
                _ck(L.cot_subsample2_forward(_p(x), _p(xs), N * Cin, H, W, BF16, st), "cot_subsample2_forward")
            else:
                xs = x[:, :, ::2, ::2].contiguous() if bp.ds_stride == 2 else x
            d0, res = new(Cout, Ho, Wo), new(Cout, Ho, Wo)
            s_d = stat(Cout, nws_o)
            _conv_bn_fwd(L, xs, None, Cin, bp.ds_conv, d0, res, bp.ds_bn, s_d, 2 * Cout, N, Cin, Cout, HWo, 0)
        else:
            xs, d0, res, s_d = None, None, x, None
        s_3 = stat(Cout, nws_o)
        ps = _drop_path_scale(blk, N, dev)  # stochastic depth: per-sample 0 or 1 / keep on the normalised branch
        m3 = _relu_mask(L, N, Cout, HWo, dev)
        _conv_bn_fwd(L, cot_out, None, Cw, bp.conv3, c3, y, bp.bn3, s_3, 2 * Cout, N, Cw, Cout, HWo, 1, residual=res, ps=ps, mask=m3)
        ctx.blk, ctx.geom, ctx.has_ds, ctx.has_ps, ctx.has_mask = blk, geom, bp.ds_conv is not None, ps is not None, m3 is not None
        extra = (x, c1, a1, s_1, cot_out, c3, y, s_3) + ((d0, s_d, xs) if bp.ds_conv is not None else ()) + \
            ((m3,) if m3 is not None else ()) + ((ps,) if ps is not None else ())
        ctx.save_for_backward(*(saved + extra))
        return y

    @staticmethod
    @_one_stream_query
    def backward(ctx, gout):
        L = _lib.lib()
        blk = ctx.blk
        bp = _block_plan(blk)
        t = ctx.saved_tensors
        saved, extra = t[:_N_SAVED], t[_N_SAVED:]
        x, c1, a1, s_1, cot_out, c3, y, s_3 = extra[:8]
        N, Cin, H, W = x.shape
        Cw, Cout = bp.conv1.out_channels, bp.conv3.out_channels
        Ho, Wo = y.shape[2], y.shape[3]
        HW, HWo = H * W, Ho * Wo
        dev, st = x.device, _stream()
        ws_a, nws_w, _ = _block_sizes(L, N, Cin, Cw, Cout, HW)
        ws_b, _, nws_o = _block_sizes(L, N, Cin, Cw, Cout, HWo)
        ws = torch.empty(max(ws_a, ws_b), dtype=torch.uint8, device=dev)
        cN, cC, cH, cW = saved[0].shape
        cpl = _plan(bp.cot)
        side = _Side(dev, max(ws_a, ws_b, _sizes(L, cN, cC, cH, cW, cpl.se0.out_channels, cpl.ke0.groups, cpl.grouped)[0]), ws,
                     bp.params)
        gout = gout.contiguous()
        # bn3 + residual + relu: dx of the normalisation and the residual's gradient in one pass
        ps = extra[-1] if ctx.has_ps else None
        m3 = extra[-2 if ctx.has_ps else -1] if ctx.has_mask else None
        # identity shortcut with a sign mask: the residual's gradient is folded into conv1's data gradient below, never written
        fold = (not ctx.has_ds) and m3 is not None and not bp.avd and _res_fold_ok(L, N, Cin, Cw, HW)
        g_c3, g_res = torch.empty_like(c3), (None if fold else torch.empty_like(c3))
        d_bn3_w, d_bn3_b = _bn_bwd(L, gout, c3, y, g_c3, bp.bn3, s_3, N, Cout, HWo, 1, nws_o, dres=g_res, ps=ps, mask=m3)
        g_cot_out = torch.empty_like(cot_out)
        _ck(L.cot_conv1x1_backward_data(_p(g_c3), _p(bp.conv3.weight), _p(g_cot_out), None, Cw, 0, _p(ws), N, Cw, Cout, HWo,
                                        BF16, st), "cot_conv1x1_backward_data")
        g_w3 = grad_sink.out_like(bp.conv3.weight)
        side.run(lambda st_, a_=(_p(g_c3), _p(cot_out), None, Cw, _p(g_w3), None, _p(side.ws), N, Cw, Cout, HWo, BF16): _ck(L.cot_conv1x1_backward_weight(*a_, st_), "cot_conv1x1_backward_weight"), g_c3, cot_out)
        if bp.avd_post:  # (cot_out at the output resolution was the pooled tensor: its gradient goes back through the blur)
            g_full = torch.empty((N, Cw, H, W), dtype=x.dtype, device=dev)
            _ck(L.cot_blurpool3x3s2_backward(_p(g_cot_out), _p(g_full), N * Cw, H, W, BF16, st), "cot_blurpool3x3s2_backward")
            g_layer_out = g_full
        else:
            g_layer_out = g_cot_out
        g_p1, g_cot = _cot_backward(L, bp.cot, saved, ctx.geom, g_layer_out, side)
        if bp.avd:
            g_a1 = torch.empty_like(a1)
            _ck(L.cot_avgpool3x3s2_backward(_p(g_p1), _p(g_a1), N * Cw, H, W, BF16, st), "cot_avgpool3x3s2_backward")
            g_c1 = torch.empty_like(c1)
        else:
            g_a1 = g_p1
            g_c1 = g_layer_out  # (reuse: consumed by the layer's backward)
        d_bn1_w, d_bn1_b = _bn_bwd(L, g_a1, c1, None, g_c1, bp.bn1, s_1, N, Cw, HW, 1, nws_w)
        g_ds = ()
        if ctx.has_ds:
            d0, s_d, xs = extra[8], extra[9], extra[10]
            # (g_c3 is still being read by conv3's weight gradient on the side stream: no reuse of its buffer here)
            g_d0 = torch.empty_like(g_c3) if side.on else g_c3
            d_ds_w, d_ds_b = _bn_bwd(L, g_res, d0, None, g_d0, bp.ds_bn, s_d, N, Cout, HWo, 0, nws_o)
            if bp.ds_pool2:  # the projection saw 2 x 2 averages: its data gradient is spread over the four pixels of each window
                g_xs = torch.empty_like(xs)
                _ck(L.cot_conv1x1_backward_data(_p(g_d0), _p(bp.ds_conv.weight), _p(g_xs), None, Cin, 0, _p(ws), N, Cin,
                                                Cout, HWo, BF16, st), "cot_conv1x1_backward_data")
                gx = torch.empty_like(x)
                _ck(L.cot_avgpool2x2s2_backward(_p(g_xs), _p(gx), N * Cin, H, W, BF16, st), "cot_avgpool2x2s2_backward")
            elif bp.ds_stride == 2:  # the projection saw every second pixel: its data gradient lands there, zeros elsewhere
                g_xs = torch.empty_like(xs)
                _ck(L.cot_conv1x1_backward_data(_p(g_d0), _p(bp.ds_conv.weight), _p(g_xs), None, Cin, 0, _p(ws), N, Cin,
                                                Cout, HWo, BF16, st), "cot_conv1x1_backward_data")
                if H % 2 == 0 and W % 2 == 0:  # values back in place and the zeros around them in one pass
                    gx = torch.empty_like(x)
                    _ck(L.cot_subsample2_backward(_p(g_xs), _p(gx), N * Cin, H, W, BF16, st), "cot_subsample2_backward")
                else:
                    gx = torch.zeros_like(x)
                    gx[:, :, ::2, ::2] = g_xs
            else:
                gx = torch.empty_like(x)
                _ck(L.cot_conv1x1_backward_data(_p(g_d0), _p(bp.ds_conv.weight), _p(gx), None, Cin, 0, _p(ws), N, Cin,
                                                Cout, HWo, BF16, st), "cot_conv1x1_backward_data")
            g_wd = grad_sink.out_like(bp.ds_conv.weight)
            side.run(lambda st_, a_=(_p(g_d0), _p(xs), None, Cin, _p(g_wd), None, _p(side.ws), N, Cin, Cout, HWo, BF16): _ck(L.cot_conv1x1_backward_weight(*a_, st_), "cot_conv1x1_backward_weight"), g_d0, xs)
            g_ds = (g_wd, d_ds_w, d_ds_b)
        else:
            gx = g_res  # identity shortcut: the residual's gradient is the first contribution to dx
        g_w1 = grad_sink.out_like(bp.conv1.weight)  # (issued before its data gradient: the two overlap)
        side.run(lambda st_, a_=(_p(g_c1), _p(x), None, Cin, _p(g_w1), None, _p(side.ws), N, Cin, Cw, HW, BF16): _ck(L.cot_conv1x1_backward_weight(*a_, st_), "cot_conv1x1_backward_weight"), g_c1, x)
        if fold:
            gx = torch.empty_like(x)
            _ck(L.cot_conv1x1_backward_data_relu_res(_p(g_c1), _p(bp.conv1.weight), _p(gx), _p(gout), _p(m3), N, Cin, Cw, HW, BF16, st),
                "cot_conv1x1_backward_data_relu_res")
        else:
            _ck(L.cot_conv1x1_backward_data(_p(g_c1), _p(bp.conv1.weight), _p(gx), None, Cin, 1, _p(ws), N, Cin, Cw, HW, BF16,
                                            st), "cot_conv1x1_backward_data")
        side.join()
        return (None, gx, g_w1, d_bn1_w, d_bn1_b) + g_cot + (g_w3, d_bn3_w, d_bn3_b) + g_ds


def block_eligible(blk, x):
    """training-mode cotnet.Bottleneck (no avd pooling, no drop-block/path) whose CotLayer is eligible"""
    if not (ENABLED and blk.training and (x.is_cuda or not _DEVICE_ONLY) and x.dim() == 4
            and x.dtype == torch.bfloat16 and x.is_contiguous() and x.data_ptr() % 16 == 0):
        return False
    bp = _block_plan(blk)
    if not (bp.static_ok and x.shape[1] == bp.conv1.in_channels and bp.conv1.weight.dtype == torch.bfloat16
            and bp.conv3.weight.dtype == torch.bfloat16 and bp.bn1.weight.dtype == torch.float32 and bp.bn1.training
            and (bp.ds_conv is not None or (bp.conv1.in_channels == bp.conv3.out_channels and not bp.avd))
            and not (bp.ds_pool2 and (x.shape[2] % 2 or x.shape[3] % 2))):  # (cot_avgpool2x2s2_*: even planes)
        return False
    pl = _plan(bp.cot)
    return (pl.ke0.weight.dtype == torch.bfloat16 and pl.em3.weight.dtype == torch.bfloat16
            and pl.gn.weight.dtype == torch.bfloat16 and pl.bn.weight.dtype == torch.float32 and pl.bn.training)


def block_forward(blk, x):
    NODE_COUNTS["bottleneck"] += 1
    return _BottleneckNode.apply(blk, x, *_block_plan(blk).params)


# ---- the other node kinds live in their own modules (round 6 split): SplitAttn blocks, channel-major Bottlenecks, the
# eval-mode call sequence.  They import this module's helpers, so they are imported HERE, at its end, and re-exported under the names
# the wrappers, bench.py and the tests have always used.  The channel-major switches are attributes of THIS module (tests rebind them).
CM_LAYOUT = os.environ.get("COT_CM_LAYOUT", "1") != "0"
CM_OPENING = os.environ.get("COT_CM_OPENING", "1") != "0"  # the stage's stride-2 opening block on the channel-major node too (A/B switch)
GX_SLABS = os.environ.get("COT_GX_SLABS", "1") != "0"  # CoXtLayer.embed[0] in channel-major blocks: two-slab kernels per group instead of torch.stack (cot_block_cm.py)
from .cot_block_sa import (_SABlockPlan, _SA_PLANS, _SASIZES, _sa_plan, _sa_sizes, _SplitAttnBlockNode, sa_block_eligible,  # noqa: E402,F401
                           sa_block_forward)
from .cot_block_cm import (_CM_OK, _CM_SIZES, _BottleneckCMNode, _bn_bwd_lay, _bn_fwd_lay, _cm_buf, _cm_geometry_ok, _cm_sizes,  # noqa: E402,F401
                           _cm_static_ok, _cm_view, _is_cm, cm_block_eligible, cm_block_forward, plan_stage_layouts)
from .cot_block_eval import _bn_inf, eval_block_eligible, eval_block_forward  # noqa: E402,F401

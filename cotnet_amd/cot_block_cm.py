"""Channel-major Bottlenecks of the 14 x 14 / 7 x 7 stages as ONE autograd node (split out of cot_layer_fused.py in round 6).
The switches (CM_LAYOUT, CM_OPENING, ENABLED, ...) stay attributes of cot_layer_fused and are read as clf.NAME at call time.
Imported by cot_layer_fused at its end: import THAT module.
"""
import ctypes
import torch
from torch.autograd import Function
from . import _lib, grad_sink
from . import cot_layer_fused as clf
from .cot_layer_fused import (  # noqa: E402  (helpers; the switches are read as clf.NAME at call time: tests rebind them there)
    BF16, NODE_COUNTS, _Side, _block_plan, _bn_bwd, _bn_fwd, _ck, _conv3x3_dgrad, _conv3x3_fwd, _drop_path_scale,
    _guard_elems, _masks, _new_guarded, _one_stream_query, _p, _plan, _relu_mask, _stream)

# ---- channel-major Bottlenecks for the 14 x 14 / 7 x 7 stages (round 5).  The layers of these stages spend their time in
# 1x1 convolutions and BatchNorms whose NCHW operands are N short rows per channel (392 / 98 bytes); stored channel-major --
# [C][N][HW], a channel = ONE row of N*HW elements -- the SAME kernels, called with N = 1 and HW' = N*HW, run 1.2-1.5x faster
# (profiles/r05_probe_cnhw.log).  So an identity-shortcut block keeps every operand of a 1x1 convolution channel-major and every
# operand of a plane kernel (grouped 3x3, aggregation) NCHW; the layout changes inside the BatchNorm / GroupNorm / radix kernels that
# sit between the two kinds anyway (cot_*_lay: one layout bit per tensor):
#     x -conv1-> c1 -bn1+relu-> a1 (NCHW, for the 3x3) + a1c (channel-major, for the 1x1s)
#     a1 -3x3-> k_pre (NCHW) -bn+relu-> k (cm);  [a1c | k] -1x1-> e0 -bn+relu-> e1 -1x1-> e3 (all cm) -GroupNorm-> w (NCHW)
#     a1c -1x1-> v_pre (cm) -bn-> v (NCHW);  aggregation(v, w) -> a -bn+silu-> y (NCHW);  radix tail(y, k) -> out (cm)
#     out -conv3-> c3 (cm) -bn3 + residual + relu-> block output (cm when the next block is one of these, else NCHW)
# The block's input / output tensor objects stay honest: a channel-major activation is a [N, C, H, W] tensor whose strides are
# (HW, N*HW, W, 1) -- any torch consumer computes the right thing with it.  Same parameters, buffers and state_dict as the module
# (models/cotnet.py:181-264).  COT_CM_LAYOUT=0 opts out.
_CM_SIZES = _lib.register_cache({})
_CM_OK = _lib.register_cache({})


def _is_cm(t):
    """[N, C, H, W] tensor stored [C][N][H][W] (dense)"""
    N, C, H, W = t.shape
    return N > 1 and C > 1 and t.stride() == (H * W, N * H * W, W, 1)


def _cm_view(buf):
    """dense [C, N, H, W] buffer -> the honest [N, C, H, W] tensor over it"""
    return buf.permute(1, 0, 2, 3)


def _cm_buf(t):
    """the dense [C, N, H, W] buffer under a channel-major [N, C, H, W] tensor"""
    return t.permute(1, 0, 2, 3)


def plan_stage_layouts(stage):
    """mark, once per stage, the blocks whose successor can take a channel-major input (so that they write one)"""
    blocks = list(stage.children())
    # the plan depends on both layout switches and on WHICH blocks the stage holds (a replaced block re-plans)
    key = (clf.CM_LAYOUT, clf.CM_OPENING, tuple(id(b) for b in blocks), tuple(b.training for b in blocks))
    if getattr(stage, "_cm_planned", None) == key:
        return
    for i, b in enumerate(blocks):
        nxt = blocks[i + 1] if i + 1 < len(blocks) else None
        # (an opening block takes NCHW: its conv1 / bn1 run at the input resolution, off the channel-resident kernels)
        # (... and only a successor in training mode runs the channel-major node: an eval-mode block inside a training stage would
        # take the module path on a strided tensor -- correct, but a silent performance cliff)
        b._next_cm = bool(clf.CM_LAYOUT and nxt is not None and nxt.training and _cm_static_ok(nxt) and not _block_plan(nxt).avd)
    stage._cm_planned = key


def _cm_static_ok(blk):
    from .cotnet import Bottleneck
    if not isinstance(blk, Bottleneck):
        return False
    bp = _block_plan(blk)
    identity = bp.ds_conv is None and not bp.avd and bp.conv1.in_channels == bp.conv3.out_channels
    # the stage's opening block (models/cotnet.py:228-264 with `avd` pooling in front of the layer and a stride-2 projection): conv1 /
    # bn1 / the pooling stay NCHW at the input resolution, the layer, conv3, bn3 and the projection's BatchNorm go channel-major
    opening = clf.CM_OPENING and bp.ds_conv is not None and bp.avd and bp.ds_stride == 2
    return bp.static_ok and (identity or opening)


def _cm_sizes(L, N, Cin, C, A, G, H, W, grouped=False, Cout=None):
    """Cin / Cout: the block's input / output channels (Cout given = the stage's opening block: conv1 on 2H x 2W NCHW planes, the
    projection on the sub-sampled input)"""
    k = (N, Cin, C, A, G, H, W, grouped, Cout)
    v = _CM_SIZES.get(k)
    if v is None:
        HW, M = H * W, N * H * W
        gws = max(int(L.cot_convg_workspace(1, 2 * C, C // 2, 2, M, 1, 1)), int(L.cot_convg_workspace(1, C // 2, 9 * C // 8, 2, M, 1, 1)),
                  int(L.cot_convg_workspace(1, C, C, 2, M, 1, 1))) if grouped else 0
        ws = max(gws, int(L.cot_conv3x3g_workspace(N, C, C, G, H, W)), int(L.cot_conv1x1_workspace(1, 2 * C, C // 2, M, 0)),
                 int(L.cot_conv1x1_workspace(1, C // 2, 9 * C // 8, M, 1)), int(L.cot_conv1x1_workspace(1, C, C, M, 0)),
                 int(L.cot_conv1x1_workspace(1, C, A, N, 1)), int(L.cot_conv1x1_workspace(1, A, 2 * C, N, 1)),
                 int(L.cot_conv1x1_workspace(1, Cin, C, M, 0)), int(L.cot_conv1x1_workspace(1, C, Cout or Cin, M, 0)),
                 int(L.cot_conv1x1_workspace(N, Cin, C, HW, 0)),
                 *((int(L.cot_conv1x1_workspace(N, Cin, C, 4 * HW, 0)), int(L.cot_conv1x1_workspace(N, Cin, Cout, HW, 0))) if Cout else ()))
        v = _CM_SIZES[k] = (ws, int(L.cot_bn_act_workspace(N, C)), int(L.cot_bn_act_workspace(1, C)), int(L.cot_bn_act_workspace(1, C // 2)),
                            int(L.cot_bn_act_workspace(1, A)), int(L.cot_bn_act_workspace(1, Cout or Cin)))
    return v


def _gx_slabs_ok(L, C, Ch, M):
    """CoXtLayer.embed[0] as two two-slab 1x1 convolutions (one per group) on channel-major operands.  _cm_geometry_ok's C % 64 == 0 makes
    every slab (C/2 rows of x or k, Ch/2 = C/4 output rows) 16-byte aligned with a reduction count the 1x1 kernels take; CoTNeXt's widths
    (C = 384 / 768 at 14x14 / 7x7) land on the LDS kernels, narrower ones on the first-generation kernel"""
    return clf.GX_SLABS and C % 64 == 0 and Ch * 2 == C


def _cm_geometry_ok(L, N, Cin, C, H, W, grouped=False):
    k = (N, Cin, C, H, W, grouped)
    v = _CM_OK.get(k)
    if v is None:
        HW = H * W
        v = bool(HW <= 256 and (N * HW) % 8 == 0 and N > 1 and C % 64 == 0
                 and L.cot_bn_act_lay_covers(N, C, HW, BF16) and L.cot_bn_act_lay_covers(N, Cin, HW, BF16)
                 and L.cot_conv1x1_lds_covers(Cin, Cin, 0, N * HW) and L.cot_conv1x1_lds_covers(C, C, 0, N * HW))
        if v and not grouped:
            v = bool(L.cot_conv1x1_lds_covers(2 * C, C, 1, N * HW) and L.cot_conv1x1_lds_covers(C // 2, C // 2, 0, N * HW))
        _CM_OK[k] = v  # (CoXtLayer's grouped 1x1s: cot_conv1x1g_* routes each group to the tuned or the general kernels itself)
    return v


def cm_block_eligible(blk, x):
    """training-mode identity-shortcut cotnet.Bottleneck of a deep stage on a bf16 tensor that is NCHW-contiguous or channel-major"""
    if not (clf.ENABLED and clf.CM_LAYOUT and blk.training and (x.is_cuda or not clf._DEVICE_ONLY) and x.dim() == 4
            and x.dtype == torch.bfloat16 and x.data_ptr() % 16 == 0 and (x.is_contiguous() or _is_cm(x))):
        return False
    if not _cm_static_ok(blk):
        return False
    bp = _block_plan(blk)
    pl = _plan(bp.cot)
    N, Cin, H, W = x.shape
    if bp.avd:
        if not (x.is_contiguous() and H % 2 == 0 and W % 2 == 0 and bp.ds_bn.training):
            return False
        H, W = H // 2, W // 2
    return (x.shape[1] == bp.conv1.in_channels and bp.conv1.weight.dtype == torch.bfloat16 and bp.conv3.weight.dtype == torch.bfloat16
            and bp.bn1.weight.dtype == torch.float32 and bp.bn1.training and bp.bn3.training
            and pl.ke0.weight.dtype == torch.bfloat16 and pl.em3.weight.dtype == torch.bfloat16 and pl.gn.weight.dtype == torch.bfloat16
            and pl.bn.weight.dtype == torch.float32 and pl.bn.training and pl.ke1.training
            and _cm_geometry_ok(_lib.lib(), N, bp.conv3.out_channels, bp.conv1.out_channels, H, W, pl.grouped))


def cm_block_forward(blk, x):
    NODE_COUNTS["bottleneck_channel_major"] += 1
    return _BottleneckCMNode.apply(blk, x, *_block_plan(blk).params)


def _bn_fwd_lay(L, x, res, y, y2, bn, stats, N, C, HW, act, lay, ps=None):
    _ck(L.cot_bn_act_forward_lay(_p(x), _p(res), _p(y), _p(y2), _p(bn.weight), _p(bn.bias), _p(stats), _p(stats[C:]),
                                 _p(bn.running_mean), _p(bn.running_var), _p(bn.num_batches_tracked), _p(ps), N, C, HW,
                                 float(bn.eps), float(bn.momentum), act, lay, BF16, _stream()), "cot_bn_act_forward_lay")


def _bn_bwd_lay(L, dy, dy2, x, y, dx, dres, bn, stats, N, C, HW, act, lay, ps=None):
    dg, db = grad_sink.out_like(bn.weight), grad_sink.out_like(bn.bias)
    _ck(L.cot_bn_act_backward_lay(_p(dy), _p(dy2), _p(x), _p(y), _p(dx), _p(dres), _p(bn.weight), _p(bn.bias), _p(stats), _p(stats[C:]),
                                  _p(dg), _p(db), _p(ps), N, C, HW, act, lay, BF16, _stream()), "cot_bn_act_backward_lay")
    return dg, db


class _BottleneckCMNode(Function):
    @staticmethod
    @_one_stream_query
    def forward(ctx, blk, x, *params):
        L = _lib.lib()
        bp = _block_plan(blk)
        pl = _plan(bp.cot)
        N, Cin, H0, W0 = x.shape
        opening = bp.avd   # the stage's first block: 3x3/2 average pooling in front of the layer, stride-2 projection shortcut
        H, W = (H0 // 2, W0 // 2) if opening else (H0, W0)
        C, A, G, Cout = bp.conv1.out_channels, pl.se0.out_channels, pl.ke0.groups, bp.conv3.out_channels
        HW, M, Ch, Ce = H * W, N * H * W, C // 2, 9 * C // 8
        dev, st = x.device, _stream()
        in_cm = _is_cm(x)
        out_cm = bool(getattr(blk, "_next_cm", False))
        GX = pl.grouped  # CoXtLayer (models/cotnet.py:106-178): grouped 1x1s (a group = a contiguous range of channel ROWS here), [x, k]
        #                  interleaved row by row, the two groups folded into the batch for the aggregation (views of the NCHW tensors)
        ws_bytes, nws_c, nws_c1, nws_h1, nws_a, nws_o1 = _cm_sizes(L, N, Cin, C, A, G, H, W, GX, Cout if opening else None)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
        masks = _masks(L, H, W, dev)
        nchw = lambda c: torch.empty((N, c, H, W), dtype=x.dtype, device=dev)  # noqa: E731
        cmj = lambda c: torch.empty((c, N, H, W), dtype=x.dtype, device=dev)   # noqa: E731  (dense channel-major buffers)
        stat = lambda c, nws: torch.empty(2 * c + nws, dtype=torch.float32, device=dev)  # noqa: E731
        xb = _cm_buf(x) if in_cm else x   # the dense buffer under x, either layout

        if opening:
            # conv1 -> bn1 + relu on the 2H x 2W input planes (NCHW, the ordinary kernels), pooled to H x W: a1 (NCHW with margins) and
            # its channel-major copy for the 1x1 convolutions
            c1 = torch.empty((N, C, H0, W0), dtype=x.dtype, device=dev)
            _ck(L.cot_conv1x1_forward(_p(xb), None, Cin, _p(bp.conv1.weight), None, _p(c1), N, Cin, C, H0 * W0, BF16, st), "cot_conv1x1_forward")
            a1f = torch.empty_like(c1)
            s_1 = stat(C, nws_c)
            _bn_fwd(L, c1, a1f, bp.bn1, s_1, 2 * C, N, C, H0 * W0, 1)
            a1 = _new_guarded(N, C, H, W, x.dtype, dev)
            _ck(L.cot_avgpool3x3s2_forward(_p(a1f), _p(a1), N * C, H0, W0, BF16, st), "cot_avgpool3x3s2_forward")
            a1c = a1.permute(1, 0, 2, 3).contiguous()
        else:
            # conv1 -> bn1 + relu -> a1 (NCHW with margins: the 3x3 weight gradient reads it shifted) and a1c (channel-major)
            c1 = cmj(C) if in_cm else nchw(C)
            if in_cm:
                _ck(L.cot_conv1x1_forward(_p(xb), None, Cin, _p(bp.conv1.weight), None, _p(c1), 1, Cin, C, M, BF16, st), "cot_conv1x1_forward")
            else:
                _ck(L.cot_conv1x1_forward(_p(xb), None, Cin, _p(bp.conv1.weight), None, _p(c1), N, Cin, C, HW, BF16, st), "cot_conv1x1_forward")
            a1, a1c = _new_guarded(N, C, H, W, x.dtype, dev), cmj(C)
            s_1 = stat(C, 0)
            _bn_fwd_lay(L, c1, None, a1, a1c, bp.bn1, s_1, N, C, HW, 1, (1 if in_cm else 0) | 8)
        # static context: grouped 3x3 (NCHW) -> bn + relu -> k (channel-major)                                   (ref :80)
        k_pre, k = nchw(C), cmj(C)
        _conv3x3_fwd(L, pl.ke0, a1, k_pre, masks, ws, N, C, G, H, W)
        s_k = stat(C, 0)
        _bn_fwd_lay(L, k_pre, None, k, None, pl.ke1, s_k, N, C, HW, 1, 4)
        # attention logits from [x | k]: two 1x1 convolutions on channel rows, GroupNorm writes the aggregation's weights NCHW (ref :81-85)
        e0, e1, e3 = cmj(Ch), cmj(Ch), cmj(Ce)
        qk = None
        if GX and _gx_slabs_ok(L, C, Ch, M):
            # CoXtLayer's embed[0] reads the row-INTERLEAVED [x0, k0, x1, k1, ...] in two groups (ref :153-154): group g sees the rows
            # g*C/2 .. of x and of k.  Channel-major, each of those is one contiguous slab, so a group is the two-slab 1x1 kernel on [x_g | k_g]
            # with the group's weight columns de-interleaved to match -- a copy of the (small) weight per step instead of a copy of the
            # activations (torch.stack: 2 C*M elements written and read) and, backward, two strided adds of C*M elements each
            qk = pl.em0.weight.view(Ch, C // 2, 2).permute(0, 2, 1).reshape(Ch, C)  # [Ch][x-part C/2 | k-part C/2], a copy (kept for the backward)
            Hc, Mg = C // 2, Ch // 2
            for g_ in range(2):
                _ck(L.cot_conv1x1_forward(_p(a1c[g_ * Hc:]), _p(k[g_ * Hc:]), Hc, _p(qk[g_ * Mg:]), None, _p(e0[g_ * Mg:]), 1, C, Mg, M, BF16, st),
                    "cot_conv1x1_forward")
        elif GX:
            qk = torch.stack([a1c, k], dim=1).view(2 * C, N, H, W)  # rows x0, k0, x1, k1, ... (ref :153-154)
            _ck(L.cot_conv1x1g_forward(_p(qk), _p(pl.em0.weight), None, _p(e0), 1, 2 * C, Ch, 2, M, BF16, st), "cot_conv1x1g_forward")
        else:
            _ck(L.cot_conv1x1_forward(_p(a1c), _p(k), C, _p(pl.em0.weight), None, _p(e0), 1, 2 * C, Ch, M, BF16, st), "cot_conv1x1_forward")
        s_e = stat(Ch, nws_h1)
        _bn_fwd(L, e0, e1, pl.em1, s_e, 2 * Ch, 1, Ch, M, 1)
        if GX:
            _ck(L.cot_conv1x1g_forward(_p(e1), _p(pl.em3.weight), _p(pl.em3.bias), _p(e3), 1, Ch, Ce, 2, M, BF16, st), "cot_conv1x1g_forward")
        else:
            _ck(L.cot_conv1x1_forward(_p(e1), None, Ch, _p(pl.em3.weight), _p(pl.em3.bias), _p(e3), 1, Ch, Ce, M, BF16, st),
                "cot_conv1x1_forward")
        gn = pl.gn
        w = nchw(Ce)
        gn_mean = torch.empty(2 * N * gn.num_groups, dtype=torch.float32, device=dev)
        gn_rstd = gn_mean[N * gn.num_groups:]
        _ck(L.cot_group_norm9_forward_lay(_p(e3), _p(gn.weight), _p(gn.bias), _p(w), _p(gn_mean), _p(gn_rstd), N, Ce, HW, float(gn.eps), 1,
                                          BF16, st), "cot_group_norm9_forward_lay")
        # values: 1x1 on channel rows, its BatchNorm writes NCHW                                                     (ref :87)
        v_pre, v = cmj(C), nchw(C)
        if GX:
            _ck(L.cot_conv1x1g_forward(_p(a1c), _p(pl.cv0.weight), None, _p(v_pre), 1, C, C, 2, M, BF16, st), "cot_conv1x1g_forward")
        else:
            _ck(L.cot_conv1x1_forward(_p(a1c), None, C, _p(pl.cv0.weight), None, _p(v_pre), 1, C, C, M, BF16, st), "cot_conv1x1_forward")
        s_v = stat(C, 0)
        _bn_fwd_lay(L, v_pre, None, v, None, pl.cv1, s_v, N, C, HW, 0, 1)
        # aggregation, bn + swish (NCHW)                                                                            (ref :88-90)
        geom = _lib.AggGeom(2 * N, C // 2, H, W, 1, C // 16, 3, 3, 1, 1, 1, 1, 1, 1) if GX else \
            _lib.AggGeom(N, C, H, W, 1, C // 8, 3, 3, 1, 1, 1, 1, 1, 1)
        bn_tail = clf.BN_TAIL  # BatchNorm + SiLU folded into the radix tail (cot_radix_*_bn): y = silu(bn(a)) is never written
        a, y = nchw(C), (None if bn_tail else nchw(C))
        bnl = pl.bn
        s_y = stat(C, nws_c)
        if bn_tail:  # (aggregation + the statistics of bn out of its epilogue; bn + swish themselves happen inside the tail's kernels)
            y_final = clf._agg_fwd_stats(L, v, w, a, None, None, None, geom, bnl, s_y, N, C, H, W)
        else:
            _ck(L.cot_agg_forward(_p(v), _p(w), _p(a), ctypes.byref(geom), BF16, _lib.COT_NCHW, st), "cot_agg_forward")
            _bn_fwd(L, a, y, bnl, s_y, 2 * C, N, C, HW, 2)
        # radix-2 split attention: y NCHW, k channel-major, the mix written channel-major for conv3                (ref :92-104)
        row = lambda c: torch.empty((c, N), dtype=x.dtype, device=dev)  # noqa: E731
        gapT, hpre, h, logitsT = row(C), row(A), row(A), row(2 * C)
        if bn_tail:
            clf._tail_gap(L, a, k, gapT, bnl, s_y, y_final, N, C, HW, 2)
        else:
            _ck(L.cot_radix_gap_t_lay(_p(y), _p(k), _p(gapT), N, C, HW, 2, BF16, st), "cot_radix_gap_t_lay")
        _ck(L.cot_conv1x1_forward(_p(gapT), None, C, _p(pl.se0.weight), _p(pl.se0.bias), _p(hpre), 1, C, A, N, BF16, st), "cot_conv1x1_forward")
        s_a = stat(A, nws_a)
        _bn_fwd(L, hpre, h, pl.sebn, s_a, 2 * A, 1, A, N, 1)
        _ck(L.cot_conv1x1_forward(_p(h), None, A, _p(pl.se3.weight), _p(pl.se3.bias), _p(logitsT), 1, A, 2 * C, N, BF16, st),
            "cot_conv1x1_forward")
        attn = torch.empty((N, C, 2), dtype=x.dtype, device=dev)
        cot_out = cmj(C)
        if bn_tail:
            _ck(L.cot_radix_mix_logits_bn(_p(a), _p(k), _p(logitsT), _p(cot_out), _p(attn), _p(bnl.weight), _p(bnl.bias), _p(s_y), _p(s_y[C:]),
                                          N, C, HW, 2 | 4, BF16, st), "cot_radix_mix_logits_bn")
        else:
            _ck(L.cot_radix_mix_logits_lay(_p(y), _p(k), _p(logitsT), _p(cot_out), _p(attn), N, C, HW, 2 | 4, BF16, st), "cot_radix_mix_logits_lay")
        # conv3 -> bn3 + residual + relu.  The residual: the block's input (identity) or bn(conv1x1(every second pixel of it))
        c3 = cmj(Cout)
        _ck(L.cot_conv1x1_forward(_p(cot_out), None, C, _p(bp.conv3.weight), None, _p(c3), 1, C, Cout, M, BF16, st), "cot_conv1x1_forward")
        ps = _drop_path_scale(blk, N, dev)
        yb = cmj(Cout) if out_cm else nchw(Cout)
        if opening:
            xs = torch.empty((N, Cin, H, W), dtype=x.dtype, device=dev)
            _ck(L.cot_subsample2_forward(_p(xb), _p(xs), N * Cin, H0, W0, BF16, st), "cot_subsample2_forward")
            d0 = nchw(Cout)
            _ck(L.cot_conv1x1_forward(_p(xs), None, Cin, _p(bp.ds_conv.weight), None, _p(d0), N, Cin, Cout, HW, BF16, st), "cot_conv1x1_forward")
            res_cm = out_cm  # (the projection's BatchNorm writes the layout bn3 writes)
            res = cmj(Cout) if res_cm else nchw(Cout)
            s_d = stat(Cout, 0)
            _bn_fwd_lay(L, d0, None, res, None, bp.ds_bn, s_d, N, Cout, HW, 0, 4 if res_cm else 0)
        else:
            xs = d0 = s_d = None
            res, res_cm = xb, in_cm
        m3 = None
        if res_cm and out_cm and ps is None:  # everything channel-major: the ordinary kernels on N*HW-element rows (16-byte accesses, sign mask)
            s_3 = stat(Cout, nws_o1)
            m3 = _relu_mask(L, 1, Cout, M, dev)
            _bn_fwd(L, c3, yb, bp.bn3, s_3, 2 * Cout, 1, Cout, M, 1, residual=res, mask=m3)
        else:
            s_3 = stat(Cout, 0)
            _bn_fwd_lay(L, c3, res, yb, None, bp.bn3, s_3, N, Cout, HW, 1, 1 | (2 if res_cm else 0) | (4 if out_cm else 0), ps=ps)
        ctx.blk, ctx.geom, ctx.flags = blk, geom, (in_cm, out_cm, ps is not None, m3 is not None, res_cm)
        ctx.save_for_backward(xb, c1, a1, a1c, s_1, k_pre, k, s_k, e0, e1, s_e, e3, w, gn_mean, gn_rstd, v_pre, v, s_v, a, y, s_y, attn,
                              gapT, hpre, h, s_a, cot_out, c3, yb, s_3, qk if GX else s_3, *((d0, s_d, xs) if opening else ()),
                              *((m3,) if m3 is not None else ()), *((ps,) if ps is not None else ()))
        return _cm_view(yb) if out_cm else yb

    @staticmethod
    @_one_stream_query
    def backward(ctx, gout):
        L = _lib.lib()
        blk = ctx.blk
        bp = _block_plan(blk)
        pl = _plan(bp.cot)
        t = ctx.saved_tensors
        (xb, c1, a1, a1c, s_1, k_pre, k, s_k, e0, e1, s_e, e3, w, gn_mean, gn_rstd, v_pre, v, s_v, a, y, s_y, attn,
         gapT, hpre, h, s_a, cot_out, c3, yb, s_3, qk) = t[:31]
        in_cm, out_cm, has_ps, has_mask, res_cm = ctx.flags
        GX = pl.grouped
        opening = bp.avd
        nx = 34 if opening else 31
        d0, s_d, xs = t[31:34] if opening else (None, None, None)
        m3 = t[nx] if has_mask else None
        ps = t[-1] if has_ps else None
        N, C, H, W = a1.shape
        Cin, Cout, A, G = bp.conv1.in_channels, c3.shape[0], pl.se0.out_channels, pl.ke0.groups
        HW, M, Ch, Ce = H * W, N * H * W, C // 2, 9 * C // 8
        H0, W0 = (xb.shape[2], xb.shape[3]) if opening else (H, W)
        dev, st = a1.device, _stream()
        ws_bytes, nws_c, nws_c1, nws_h1, nws_a, nws_o1 = _cm_sizes(L, N, Cin, C, A, G, H, W, GX, Cout if opening else None)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
        masks = _masks(L, H, W, dev)
        side = _Side(dev, ws_bytes, ws, bp.params)
        nchw = lambda c: torch.empty((N, c, H, W), dtype=a1.dtype, device=dev)  # noqa: E731
        cmj = lambda c: torch.empty((c, N, H, W), dtype=a1.dtype, device=dev)   # noqa: E731
        ke0, ke1, em0, em1, em3, cv0, cv1 = pl.ke0, pl.ke1, pl.em0, pl.em1, pl.em3, pl.cv0, pl.cv1
        se0, sebn, se3 = pl.se0, pl.sebn, pl.se3
        # the upstream gradient in the layout the forward wrote its output in
        if out_cm:
            gb = _cm_buf(gout) if _is_cm(gout) else gout.permute(1, 0, 2, 3).contiguous()
        else:
            gb = gout.contiguous()
        # bn3 + residual + relu
        g_c3 = cmj(Cout)
        # identity shortcut, everything channel-major, a sign mask: the residual's gradient is folded into conv1's data gradient below
        fold = has_mask and in_cm and not opening and clf._res_fold_ok(L, 1, Cin, C, M)
        g_res = None if fold else (cmj(Cout) if res_cm else nchw(Cout))
        if has_mask:
            d_bn3_w, d_bn3_b = _bn_bwd(L, gb, c3, None, g_c3, bp.bn3, s_3, 1, Cout, M, 1, nws_o1, dres=g_res, mask=m3)
        else:
            d_bn3_w, d_bn3_b = _bn_bwd_lay(L, gb, None, c3, yb, g_c3, g_res, bp.bn3, s_3, N, Cout, HW, 1,
                                           (1 if out_cm else 0) | 4 | (8 if out_cm else 0) | 16 | (32 if res_cm else 0), ps=ps)
        g_out = cmj(C)
        _ck(L.cot_conv1x1_backward_data(_p(g_c3), _p(bp.conv3.weight), _p(g_out), None, C, 0, _p(ws), 1, C, Cout, M, BF16, st),
            "cot_conv1x1_backward_data")
        g_w3c = grad_sink.out_like(bp.conv3.weight)
        side.run(lambda st_, a_=(_p(g_c3), _p(cot_out), None, C, _p(g_w3c), None, _p(side.ws), 1, C, Cout, M, BF16): _ck(L.cot_conv1x1_backward_weight(*a_, st_), "cot_conv1x1_backward_weight"), g_c3, cot_out)
        # radix mix -> pair-softmax backward -> se branch -> gap
        row = lambda c: torch.empty((c, N), dtype=a1.dtype, device=dev)  # noqa: E731
        glogT, gh, ggapT = row(2 * C), row(A), row(C)
        bnl = pl.bn
        if y is None:  # (the forward folded bn + swish into the tail: so does the backward)
            tsum = torch.empty(N * C * 4, dtype=torch.float32, device=dev)
            _ck(L.cot_radix_mix_backward_reduce_bn(_p(g_out), _p(a), _p(k), _p(attn), _p(glogT), _p(tsum), _p(bnl.weight), _p(bnl.bias),
                                                   _p(s_y), _p(s_y[C:]), N, C, HW, 1 | 4, BF16, st), "cot_radix_mix_backward_reduce_bn")
        else:
            _ck(L.cot_radix_mix_backward_reduce_lay(_p(g_out), _p(y), _p(k), _p(attn), _p(glogT), N, C, HW, 1 | 4, BF16, st),
                "cot_radix_mix_backward_reduce_lay")
        _ck(L.cot_conv1x1_backward_data(_p(glogT), _p(se3.weight), _p(gh), None, A, 0, _p(ws), 1, A, 2 * C, N, BF16, st), "cot_conv1x1_backward_data")
        g_w3, g_b3 = grad_sink.out_like(se3.weight), grad_sink.out_like(se3.bias)
        side.run(lambda st_, a_=(_p(glogT), _p(h), None, A, _p(g_w3), _p(g_b3), _p(side.ws), 1, A, 2 * C, N, BF16): _ck(L.cot_conv1x1_backward_weight(*a_, st_), "cot_conv1x1_backward_weight"), glogT, h)
        ghpre = row(A)
        d_sa_w, d_sa_b = _bn_bwd(L, gh, hpre, None, ghpre, sebn, s_a, 1, A, N, 1, nws_a)
        _ck(L.cot_conv1x1_backward_data(_p(ghpre), _p(se0.weight), _p(ggapT), None, C, 0, _p(ws), 1, C, A, N, BF16, st), "cot_conv1x1_backward_data")
        g_w0, g_b0 = grad_sink.out_like(se0.weight), grad_sink.out_like(se0.bias)
        side.run(lambda st_, a_=(_p(ghpre), _p(gapT), None, C, _p(g_w0), _p(g_b0), _p(side.ws), 1, C, A, N, BF16): _ck(L.cot_conv1x1_backward_weight(*a_, st_), "cot_conv1x1_backward_weight"), ghpre, gapT)
        # bn + swish, aggregation (NCHW)
        ga, gk = nchw(C), cmj(C)
        if y is None:
            d_bn_w, d_bn_b = grad_sink.out_like(bnl.weight), grad_sink.out_like(bnl.bias)
            _ck(L.cot_radix_mix_backward_apply_bn(_p(g_out), _p(a), _p(attn), _p(ggapT), _p(tsum), _p(ga), _p(gk), _p(bnl.weight),
                                                  _p(bnl.bias), _p(s_y), _p(s_y[C:]), _p(d_bn_w), _p(d_bn_b), N, C, HW, 1 | 4, BF16, st),
                "cot_radix_mix_backward_apply_bn")
        else:
            gy = nchw(C)
            _ck(L.cot_radix_mix_backward_apply_lay(_p(g_out), _p(attn), _p(ggapT), _p(gy), _p(gk), N, C, HW, 1 | 4, BF16, st),
                "cot_radix_mix_backward_apply_lay")
            d_bn_w, d_bn_b = _bn_bwd(L, gy, a, None, ga, bnl, s_y, N, C, HW, 2, nws_c)
        gv, gw = nchw(C), nchw(Ce)
        _ck(L.cot_agg_backward(_p(ga), _p(v), _p(w), _p(gv), _p(gw), ctypes.byref(ctx.geom), BF16, _lib.COT_NCHW, st), "cot_agg_backward")
        # values branch: bn (NCHW gradient in, channel-major out), 1x1 -> first contribution to the channel-major dx
        gv_pre = cmj(C)
        d_cv_w, d_cv_b = _bn_bwd_lay(L, gv, None, v_pre, None, gv_pre, None, cv1, s_v, N, C, HW, 0, 4 | 16)
        gxc = cmj(C)
        g_wv = grad_sink.out_like(cv0.weight)
        if GX:
            _ck(L.cot_conv1x1g_backward_data(_p(gv_pre), _p(cv0.weight), _p(gxc), 0, 1, C, C, 2, M, BF16, st), "cot_conv1x1g_backward_data")
            side.run(lambda st_, a_=(_p(gv_pre), _p(a1c), _p(g_wv), None, _p(side.ws), 1, C, C, 2, M, BF16): _ck(L.cot_conv1x1g_backward_weight(*a_, st_), "cot_conv1x1g_backward_weight"), gv_pre, a1c)
        else:
            _ck(L.cot_conv1x1_backward_data(_p(gv_pre), _p(cv0.weight), _p(gxc), None, C, 0, _p(ws), 1, C, C, M, BF16, st), "cot_conv1x1_backward_data")
            side.run(lambda st_, a_=(_p(gv_pre), _p(a1c), None, C, _p(g_wv), None, _p(side.ws), 1, C, C, M, BF16): _ck(L.cot_conv1x1_backward_weight(*a_, st_), "cot_conv1x1_backward_weight"), gv_pre, a1c)
        # logits branch: GroupNorm (NCHW gradient in, channel-major out), 1x1 (+bias), bn + relu, 1x1 on [x | k] -> dx +=, dk +=
        gn = pl.gn
        ge3, g_gn_w, g_gn_b = cmj(Ce), grad_sink.out_like(gn.weight), grad_sink.out_like(gn.bias)
        gn_ws = torch.empty(2 * N * Ce, dtype=torch.float32, device=dev)
        _ck(L.cot_group_norm9_backward_lay(_p(gw), _p(e3), _p(gn_mean), _p(gn_rstd), _p(gn.weight), _p(ge3), None, None, _p(gn_ws),
                                           N, Ce, HW, 2 | 4, BF16, st), "cot_group_norm9_backward_lay")
        side.run(lambda st_, a_=(_p(gn_ws), _p(g_gn_w), _p(g_gn_b), N, Ce, BF16): _ck(L.cot_group_norm9_backward_params(*a_, st_), "cot_group_norm9_backward_params"), gn_ws)
        ge1 = cmj(Ch)
        g_we3, g_be3 = grad_sink.out_like(em3.weight), grad_sink.out_like(em3.bias)
        if GX:
            _ck(L.cot_conv1x1g_backward_data(_p(ge3), _p(em3.weight), _p(ge1), 0, 1, Ch, Ce, 2, M, BF16, st), "cot_conv1x1g_backward_data")
            side.run(lambda st_, a_=(_p(ge3), _p(e1), _p(g_we3), _p(g_be3), _p(side.ws), 1, Ch, Ce, 2, M, BF16): _ck(L.cot_conv1x1g_backward_weight(*a_, st_), "cot_conv1x1g_backward_weight"), ge3, e1)
        else:
            _ck(L.cot_conv1x1_backward_data(_p(ge3), _p(em3.weight), _p(ge1), None, Ch, 0, _p(ws), 1, Ch, Ce, M, BF16, st), "cot_conv1x1_backward_data")
            side.run(lambda st_, a_=(_p(ge3), _p(e1), None, Ch, _p(g_we3), _p(g_be3), _p(side.ws), 1, Ch, Ce, M, BF16): _ck(L.cot_conv1x1_backward_weight(*a_, st_), "cot_conv1x1_backward_weight"), ge3, e1)
        ge0 = cmj(Ch)
        d_em_w, d_em_b = _bn_bwd(L, ge1, e0, None, ge0, em1, s_e, 1, Ch, M, 1, nws_h1)
        g_we0 = grad_sink.out_like(em0.weight)
        if GX and qk.dim() == 2:  # (the forward took the two-slab form: qk holds the de-interleaved weight)
            Hc, Mg = C // 2, Ch // 2
            gwp = torch.empty_like(qk)  # gradient w.r.t. the de-interleaved weight, re-interleaved into the parameter's slot below
            for g_ in range(2):
                _ck(L.cot_conv1x1_backward_data(_p(ge0[g_ * Mg:]), _p(qk[g_ * Mg:]), _p(gxc[g_ * Hc:]), _p(gk[g_ * Hc:]), Hc, 3, _p(ws), 1, C, Mg, M,
                                                BF16, st), "cot_conv1x1_backward_data")
                side.run(lambda st_, a_=(_p(ge0[g_ * Mg:]), _p(a1c[g_ * Hc:]), _p(k[g_ * Hc:]), Hc, _p(gwp[g_ * Mg:]), None, _p(side.ws), 1, C, Mg, M, BF16): _ck(L.cot_conv1x1_backward_weight(*a_, st_), "cot_conv1x1_backward_weight"), ge0, a1c, k, gwp)

            def _interleave(st_, dst=g_we0, src=gwp, side_=side):  # [Ch][2][C/2] -> [Ch][C/2][2], behind the two launches above on their stream
                if side_.on:
                    with torch.cuda.stream(side_.stream):
                        dst.view(Ch, C // 2, 2).copy_(src.view(Ch, 2, C // 2).permute(0, 2, 1))
                else:
                    dst.view(Ch, C // 2, 2).copy_(src.view(Ch, 2, C // 2).permute(0, 2, 1))
            side.run(_interleave, gwp)
        elif GX:  # gradient of the row-interleaved [x0, k0, x1, k1, ...]: de-interleaved into dx / dk (two strided adds)
            gqk = torch.empty_like(qk)
            _ck(L.cot_conv1x1g_backward_data(_p(ge0), _p(em0.weight), _p(gqk), 0, 1, 2 * C, Ch, 2, M, BF16, st), "cot_conv1x1g_backward_data")
            gq5 = gqk.view(C, 2, N, H, W)
            gxc.add_(gq5[:, 0])
            gk.add_(gq5[:, 1])
            side.run(lambda st_, a_=(_p(ge0), _p(qk), _p(g_we0), None, _p(side.ws), 1, 2 * C, Ch, 2, M, BF16): _ck(L.cot_conv1x1g_backward_weight(*a_, st_), "cot_conv1x1g_backward_weight"), ge0, qk)
        else:
            _ck(L.cot_conv1x1_backward_data(_p(ge0), _p(em0.weight), _p(gxc), _p(gk), C, 3, _p(ws), 1, 2 * C, Ch, M, BF16, st), "cot_conv1x1_backward_data")
            side.run(lambda st_, a_=(_p(ge0), _p(a1c), _p(k), C, _p(g_we0), None, _p(side.ws), 1, 2 * C, Ch, M, BF16): _ck(L.cot_conv1x1_backward_weight(*a_, st_), "cot_conv1x1_backward_weight"), ge0, a1c, k)
        # key branch: bn + relu (channel-major gradient in, NCHW out), grouped 3x3 -> the NCHW contribution to dx
        gk_pre = nchw(C)
        d_ke_w, d_ke_b = _bn_bwd_lay(L, gk, None, k_pre, None, gk_pre, None, ke1, s_k, N, C, HW, 1, 1)
        g_wk = grad_sink.out_like(ke0.weight)
        side.run(lambda st_, a_=(_p(gk_pre), _p(a1), _p(g_wk), _p(masks), _p(side.ws), N, C, C, G, H, W, BF16, _guard_elems(a1)): _ck(L.cot_conv3x3g_backward_weight_guarded(*a_, st_), "cot_conv3x3g_backward_weight"), gk_pre, a1, masks)
        gx3 = nchw(C)
        _conv3x3_dgrad(L, ke0, gk_pre, gx3, 0, masks, ws, N, C, G, H, W)
        g_w1 = grad_sink.out_like(bp.conv1.weight)
        g_ds = ()
        if opening:
            # the pooled activation's gradient = the 3x3's NCHW contribution + the 1x1s' channel-major one (a strided add), back through the
            # pooling, bn1 and conv1 at the input resolution; the projection: its BatchNorm takes the residual gradient in bn3's layout
            gx3.add_(_cm_view(gxc))
            g_a1f = torch.empty((N, C, H0, W0), dtype=a1.dtype, device=dev)
            _ck(L.cot_avgpool3x3s2_backward(_p(gx3), _p(g_a1f), N * C, H0, W0, BF16, st), "cot_avgpool3x3s2_backward")
            g_c1 = torch.empty_like(c1)
            d_bn1_w, d_bn1_b = _bn_bwd(L, g_a1f, c1, None, g_c1, bp.bn1, s_1, N, C, H0 * W0, 1, nws_c)
            g_d0 = nchw(Cout)
            d_ds_w, d_ds_b = _bn_bwd_lay(L, g_res, None, d0, None, g_d0, None, bp.ds_bn, s_d, N, Cout, HW, 0, 1 if res_cm else 0)
            g_xs = torch.empty_like(xs)
            _ck(L.cot_conv1x1_backward_data(_p(g_d0), _p(bp.ds_conv.weight), _p(g_xs), None, Cin, 0, _p(ws), N, Cin, Cout, HW, BF16, st),
                "cot_conv1x1_backward_data")
            gx = torch.empty_like(xb)
            _ck(L.cot_subsample2_backward(_p(g_xs), _p(gx), N * Cin, H0, W0, BF16, st), "cot_subsample2_backward")
            g_wd = grad_sink.out_like(bp.ds_conv.weight)
            side.run(lambda st_, a_=(_p(g_d0), _p(xs), None, Cin, _p(g_wd), None, _p(side.ws), N, Cin, Cout, HW, BF16): _ck(L.cot_conv1x1_backward_weight(*a_, st_), "cot_conv1x1_backward_weight"), g_d0, xs)
            g_ds = (g_wd, d_ds_w, d_ds_b)
            side.run(lambda st_, a_=(_p(g_c1), _p(xb), None, Cin, _p(g_w1), None, _p(side.ws), N, Cin, C, H0 * W0, BF16): _ck(L.cot_conv1x1_backward_weight(*a_, st_), "cot_conv1x1_backward_weight"), g_c1, xb)
            _ck(L.cot_conv1x1_backward_data(_p(g_c1), _p(bp.conv1.weight), _p(gx), None, Cin, 1, _p(ws), N, Cin, C, H0 * W0, BF16, st),
                "cot_conv1x1_backward_data")
        else:
            # bn1: the two contributions to da1 (channel-major from the 1x1s, NCHW from the 3x3) meet in its backward
            g_c1 = cmj(C) if in_cm else nchw(C)
            d_bn1_w, d_bn1_b = _bn_bwd_lay(L, gxc, gx3, c1, None, g_c1, None, bp.bn1, s_1, N, C, HW, 1, 1 | (4 if in_cm else 0) | (16 if in_cm else 0))
            gx = g_res  # identity shortcut: the residual's gradient is the first contribution to dx
            if in_cm:
                side.run(lambda st_, a_=(_p(g_c1), _p(xb), None, Cin, _p(g_w1), None, _p(side.ws), 1, Cin, C, M, BF16): _ck(L.cot_conv1x1_backward_weight(*a_, st_), "cot_conv1x1_backward_weight"), g_c1, xb)
                if fold:
                    gx = cmj(Cin)
                    _ck(L.cot_conv1x1_backward_data_relu_res(_p(g_c1), _p(bp.conv1.weight), _p(gx), _p(gb), _p(m3), 1, Cin, C, M, BF16, st),
                        "cot_conv1x1_backward_data_relu_res")
                else:
                    _ck(L.cot_conv1x1_backward_data(_p(g_c1), _p(bp.conv1.weight), _p(gx), None, Cin, 1, _p(ws), 1, Cin, C, M, BF16, st), "cot_conv1x1_backward_data")
            else:
                side.run(lambda st_, a_=(_p(g_c1), _p(xb), None, Cin, _p(g_w1), None, _p(side.ws), N, Cin, C, HW, BF16): _ck(L.cot_conv1x1_backward_weight(*a_, st_), "cot_conv1x1_backward_weight"), g_c1, xb)
                _ck(L.cot_conv1x1_backward_data(_p(g_c1), _p(bp.conv1.weight), _p(gx), None, Cin, 1, _p(ws), N, Cin, C, HW, BF16, st), "cot_conv1x1_backward_data")
        side.join()
        g_cot = (g_wk, d_ke_w, d_ke_b, g_we0, d_em_w, d_em_b, g_we3, g_be3, g_gn_w, g_gn_b, g_wv, d_cv_w, d_cv_b, d_bn_w, d_bn_b,
                 g_w0, g_b0, d_sa_w, d_sa_b, g_w3, g_b3)
        return (None, _cm_view(gx) if in_cm else gx, g_w1, d_bn1_w, d_bn1_b) + g_cot + (g_w3c, d_bn3_w, d_bn3_b) + g_ds

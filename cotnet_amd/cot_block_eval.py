"""Inference (eval mode, autograd off): a Bottleneck as one C-ABI call sequence on the running statistics (split out of
cot_layer_fused.py in round 6).  Imported by cot_layer_fused at its end: import THAT module.
"""
import ctypes
import torch
from . import _lib
from . import cot_layer_fused as clf
from .cot_layer_fused import (  # noqa: E402  (helpers; the switches are read as clf.NAME at call time: tests rebind them there)
    BF16, NODE_COUNTS, _block_plan, _ck, _gn_fused_ok, _masks, _one_stream_query, _p, _plan, _sizes, _stream)

# ---- inference (BASELINE config 2: forward only, eval mode, no autograd).  The same launch sequence as _BottleneckNode.forward with
# the BatchNorms on their running statistics (cot_bn_act_inference, in place: one pass each) and nothing kept for a backward:
# ~25 C-ABI calls per block instead of ~25 autograd-tracked module calls -- the eager forward is bound by the host, not the device
# (profiles/r04_bench_fwd.json: 6.26 ms to issue a 5.07 ms step).  Reference: models/cotnet.py:79-104, :228-264.
def _bn_inf(L, x, bn, N, C, HW, act, residual=None, out=None):
    y = x if out is None else out
    _ck(L.cot_bn_act_inference(_p(x), _p(residual), _p(y), _p(bn.weight), _p(bn.bias), _p(bn.running_mean), _p(bn.running_var), N, C, HW,
                               float(bn.eps), act, BF16, _stream()), "cot_bn_act_inference")
    return y


def eval_block_eligible(blk, x):
    """eval-mode cotnet.Bottleneck around an (ungrouped) CotLayer on a contiguous bf16 NCHW tensor, autograd off"""
    if not (clf.ENABLED and not blk.training and not torch.is_grad_enabled() and (x.is_cuda or not clf._DEVICE_ONLY) and x.dim() == 4
            and x.dtype == torch.bfloat16 and x.is_contiguous() and x.data_ptr() % 16 == 0):
        return False
    bp = _block_plan(blk)
    if not (bp.static_ok and x.shape[1] == bp.conv1.in_channels and bp.conv1.weight.dtype == torch.bfloat16
            and bp.conv3.weight.dtype == torch.bfloat16 and bp.bn1.weight.dtype == torch.float32
            and not bp.avd_post and not bp.ds_pool2  # (SE-CoTNetD's stage-opening blocks: training node only)
            and (bp.ds_conv is not None or (bp.conv1.in_channels == bp.conv3.out_channels and not bp.avd))):
        return False
    pl = _plan(bp.cot)
    # every BatchNorm of the block runs on its RUNNING statistics here (cot_bn_act_inference): each one has to be in eval mode
    # itself (a block in eval() with an inner BatchNorm put back into train() takes the module path), carry running statistics and
    # fp32 parameters
    bns = [bp.bn1, bp.bn3, pl.ke1, pl.em1, pl.cv1, pl.bn, pl.sebn] + ([bp.ds_bn] if bp.ds_conv is not None else [])
    if not all((not bn.training) and bn.running_mean is not None and bn.running_var is not None and bn.weight is not None
               and bn.weight.dtype == torch.float32 and bn.running_mean.dtype == torch.float32 for bn in bns):
        return False
    return (not pl.grouped and pl.ke0.weight.dtype == torch.bfloat16 and pl.em3.weight.dtype == torch.bfloat16
            and pl.gn.weight.dtype == torch.bfloat16 and x.shape[2] * x.shape[3] <= 8192 * 4)


@_one_stream_query
def eval_block_forward(blk, x):
    NODE_COUNTS["bottleneck_eval"] += 1
    L = _lib.lib()
    bp = _block_plan(blk)
    pl = _plan(bp.cot)
    N, Cin, H0, W0 = x.shape
    C, Cout, A, G = bp.conv1.out_channels, bp.conv3.out_channels, pl.se0.out_channels, pl.ke0.groups
    dev, st = x.device, _stream()
    HW0 = H0 * W0
    new = lambda c, h, w: torch.empty((N, c, h, w), dtype=x.dtype, device=dev)  # noqa: E731
    a1 = new(C, H0, W0)
    _ck(L.cot_conv1x1_forward(_p(x), None, Cin, _p(bp.conv1.weight), None, _p(a1), N, Cin, C, HW0, BF16, st), "cot_conv1x1_forward")
    _bn_inf(L, a1, bp.bn1, N, C, HW0, 1)
    if bp.avd:
        H, W = (H0 - 1) // 2 + 1, (W0 - 1) // 2 + 1
        p1 = new(C, H, W)
        _ck(L.cot_avgpool3x3s2_forward(_p(a1), _p(p1), N * C, H0, W0, BF16, st), "cot_avgpool3x3s2_forward")
    else:
        H, W, p1 = H0, W0, a1
    HW, Ch, Ce = H * W, C // 2, 9 * C // 8
    ws_bytes = _sizes(L, N, C, H, W, A, G, False)[0]
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    masks = _masks(L, H, W, dev)
    k = new(C, H, W)
    _ck(L.cot_conv3x3g_forward(_p(p1), _p(pl.ke0.weight), _p(k), _p(masks), _p(ws), N, C, C, G, H, W, BF16, st), "cot_conv3x3g_forward")
    _bn_inf(L, k, pl.ke1, N, C, HW, 1)
    e1 = new(Ch, H, W)
    _ck(L.cot_conv1x1_forward(_p(p1), _p(k), C, _p(pl.em0.weight), None, _p(e1), N, 2 * C, Ch, HW, BF16, st), "cot_conv1x1_forward")
    _bn_inf(L, e1, pl.em1, N, Ch, HW, 1)
    e3, gn = new(Ce, H, W), pl.gn
    gn_mean = torch.empty(2 * N * gn.num_groups, dtype=torch.float32, device=dev)
    gn_rstd = gn_mean[N * gn.num_groups:]
    v = new(C, H, W)
    geom = _lib.AggGeom(N, C, H, W, 1, C // 8, 3, 3, 1, 1, 1, 1, 1, 1)
    a = new(C, H, W)
    if clf.GN_FUSED and _gn_fused_ok(L, Ch, HW, W):
        part = torch.empty(int(L.cot_gn9_stats_floats(N, Ce, HW)), dtype=torch.float32, device=dev)
        _ck(L.cot_conv1x1_forward_gn9(_p(e1), None, Ch, _p(pl.em3.weight), _p(pl.em3.bias), _p(e3), _p(part), N, Ch, Ce, HW, BF16, st),
            "cot_conv1x1_forward_gn9")
        _ck(L.cot_gn9_stats_finalize(_p(part), _p(gn_mean), _p(gn_rstd), N, Ce, HW, float(gn.eps), st), "cot_gn9_stats_finalize")
        _ck(L.cot_conv1x1_forward(_p(p1), None, C, _p(pl.cv0.weight), None, _p(v), N, C, C, HW, BF16, st), "cot_conv1x1_forward")
        _bn_inf(L, v, pl.cv1, N, C, HW, 0)
        _ck(L.cot_agg_gn9_forward(_p(v), _p(e3), _p(gn_mean), _p(gn_rstd), _p(gn.weight), _p(gn.bias), gn.num_groups, _p(a),
                                  ctypes.byref(geom), BF16, st), "cot_agg_gn9_forward")
    else:
        _ck(L.cot_conv1x1_forward(_p(e1), None, Ch, _p(pl.em3.weight), _p(pl.em3.bias), _p(e3), N, Ch, Ce, HW, BF16, st), "cot_conv1x1_forward")
        if HW <= 8192:
            w = new(Ce, H, W)
            _ck(L.cot_group_norm9_forward(_p(e3), _p(gn.weight), _p(gn.bias), _p(w), _p(gn_mean), _p(gn_rstd), N, Ce, HW, float(gn.eps), BF16,
                                          st), "cot_group_norm9_forward")
        else:
            w = torch.nn.functional.group_norm(e3, gn.num_groups, gn.weight, gn.bias, gn.eps)
        _ck(L.cot_conv1x1_forward(_p(p1), None, C, _p(pl.cv0.weight), None, _p(v), N, C, C, HW, BF16, st), "cot_conv1x1_forward")
        _bn_inf(L, v, pl.cv1, N, C, HW, 0)
        _ck(L.cot_agg_forward(_p(v), _p(w), _p(a), ctypes.byref(geom), BF16, _lib.COT_NCHW, st), "cot_agg_forward")
    _bn_inf(L, a, pl.bn, N, C, HW, 2)
    row = lambda c: torch.empty((c, N), dtype=x.dtype, device=dev)  # noqa: E731
    gapT, h, logitsT = row(C), row(A), row(2 * C)
    _ck(L.cot_radix_gap_t(_p(a), _p(k), _p(gapT), N, C, HW, BF16, st), "cot_radix_gap_t")
    _ck(L.cot_conv1x1_forward(_p(gapT), None, C, _p(pl.se0.weight), _p(pl.se0.bias), _p(h), 1, C, A, N, BF16, st), "cot_conv1x1_forward")
    _bn_inf(L, h, pl.sebn, 1, A, N, 1)
    _ck(L.cot_conv1x1_forward(_p(h), None, A, _p(pl.se3.weight), _p(pl.se3.bias), _p(logitsT), 1, A, 2 * C, N, BF16, st), "cot_conv1x1_forward")
    attn = torch.empty((N, C, 2), dtype=x.dtype, device=dev)
    out = new(C, H, W)
    _ck(L.cot_radix_mix_logits(_p(a), _p(k), _p(logitsT), _p(out), _p(attn), N, C, HW, BF16, st), "cot_radix_mix_logits")
    y = new(Cout, H, W)
    _ck(L.cot_conv1x1_forward(_p(out), None, C, _p(bp.conv3.weight), None, _p(y), N, C, Cout, HW, BF16, st), "cot_conv1x1_forward")
    if bp.ds_conv is not None:
        if bp.ds_stride == 2 and H0 % 2 == 0 and W0 % 2 == 0:
            xs = new(Cin, H0 // 2, W0 // 2)
            _ck(L.cot_subsample2_forward(_p(x), _p(xs), N * Cin, H0, W0, BF16, st), "cot_subsample2_forward")
        else:
            xs = x[:, :, ::2, ::2].contiguous() if bp.ds_stride == 2 else x
        res = new(Cout, H, W)
        _ck(L.cot_conv1x1_forward(_p(xs), None, Cin, _p(bp.ds_conv.weight), None, _p(res), N, Cin, Cout, HW, BF16, st), "cot_conv1x1_forward")
        _bn_inf(L, res, bp.ds_bn, N, Cout, HW, 0)
    else:
        res = x
    return _bn_inf(L, y, bp.bn3, N, Cout, HW, 1, residual=res)

"""SE-CoTNetD's SplitAttn block as ONE autograd node (split out of cot_layer_fused.py in round 6; the switches, the side-stream machinery
and the BatchNorm / convolution helpers live there and are imported below).  Imported by cot_layer_fused at its end: import THAT module.
"""
import weakref
import torch
from torch import nn
from torch.autograd import Function
from . import _lib, grad_sink
from . import cot_layer_fused as clf
from .cot_layer_fused import (  # noqa: E402  (helpers; the switches are read as clf.NAME at call time: tests rebind them there)
    BF16, NODE_COUNTS, _Side, _bn_bwd, _bn_fwd, _bn_static_ok, _ck, _conv3x3_dgrad, _conv3x3_fwd, _conv_ok,
    _drop_path_scale, _guard_elems, _masks, _new_guarded, _one_stream_query, _p, _relu_mask, _stream)

# ---- SE-CoTNetD's OTHER block kind as one node: CoTBottleneck whose conv2 is SplitAttnConv2d(radix = 1) (models/cotnet_hybrid.py:
# 138-146, :172-202; models/layers/split_attn.py:62-88) -- conv1 -> bn1+relu -> dense 3x3 -> bn0+act -> SE gate x * sigmoid(fc2(act(
# bn(fc1(mean_hw x))))) -> conv3 -> bn3 + residual + relu.  29 of se_cotnetd_152's 50 blocks; with one autograd node per op they
# left the step host-bound (71.9 ms per step for 47.9 ms of kernels, profiles/r04).  Identity-shortcut blocks (no avd pooling, no
# projection: 26 of the 29) and, since round 6, the stage-opening ones: a projection shortcut ([AvgPool2d(2, 2),] 1x1 convolution,
# BatchNorm: the `avg_down` form, models/resnet.py:380-394) and BlurPool2d behind conv2 (avd_first False).  The two fc layers act on [N, C] descriptors:
# plain GEMMs (torch.addmm / matmul on the conv weights viewed as matrices), their BatchNorm over the batch on the library's
# small-batch kernel.
class _SABlockPlan:
    __slots__ = ("conv1", "bn1", "conv", "bn0", "fc1", "sbn", "fc2", "conv3", "bn3", "params", "static_ok", "act0", "act1",
                 "ds_conv", "ds_bn", "ds_pool2", "avd_post")

    def __init__(self, blk):
        from .layers import SplitAttnConv2d
        sa = blk.conv2
        self.conv1, self.bn1, self.conv3, self.bn3 = blk.conv1, blk.bn1, blk.conv3, blk.bn3
        self.static_ok = False
        self.params = []
        if not (isinstance(sa, SplitAttnConv2d) and sa.radix == 1):
            return
        self.conv, self.bn0, self.fc1, self.sbn, self.fc2 = sa.conv, sa.bn0, sa.fc1, sa.bn1, sa.fc2
        code = lambda m: 1 if isinstance(m, nn.ReLU) else (2 if isinstance(m, nn.SiLU) else -1)  # noqa: E731
        self.act0, self.act1 = code(sa.act0), code(sa.act1)
        C = sa.conv.out_channels
        fc_ok = lambda c: (isinstance(c, nn.Conv2d) and c.kernel_size == (1, 1) and c.stride == (1, 1) and c.padding == (0, 0)  # noqa: E731
                           and c.groups == 1 and c.bias is not None)
        from .layers import BlurPool2d
        ds, avd = blk.downsample, blk.avd
        self.ds_conv = self.ds_bn = None
        self.ds_pool2 = False
        self.avd_post = (avd is not None and not getattr(blk, "avd_first", True) and isinstance(avd, BlurPool2d) and avd.filt_size == 3
                         and avd.stride == 2)
        shape_ok = ds is None and avd is None
        if isinstance(ds, nn.Sequential) and len(ds) == 3 and isinstance(ds[0], (nn.Identity, nn.AvgPool2d)):
            pool = ds[0] if isinstance(ds[0], nn.AvgPool2d) else None
            self.ds_pool2 = (pool is not None and pool.kernel_size == 2 and pool.stride == 2 and pool.padding == 0
                             and pool.divisor_override is None)  # (ceil_mode / count_include_pad: no effect on even planes, checked at run time)
            c = ds[1]
            self.ds_conv, self.ds_bn = c, ds[2]
            # stride 1: no pooling anywhere; stride 2: BlurPool behind conv2 AND the 2 x 2 average in the shortcut
            shape_ok = (isinstance(c, nn.Conv2d) and c.kernel_size == (1, 1) and c.stride == (1, 1) and c.padding == (0, 0)
                        and c.groups == 1 and c.bias is None and _bn_static_ok(ds[2]) and c.in_channels == blk.conv1.in_channels
                        and c.out_channels == blk.conv3.out_channels and c.out_channels % 8 == 0
                        and ((avd is None and pool is None) or (self.avd_post and self.ds_pool2)))
        self.static_ok = (
            shape_ok and blk.drop_block is None and sa.drop_block is None
            and (blk.drop_path is None or hasattr(blk.drop_path, "drop_prob")) and getattr(blk, "se", None) is None
            and isinstance(blk.act1, nn.ReLU) and isinstance(blk.act3, nn.ReLU)
            and _conv_ok(blk.conv1, 1, 1) and blk.conv1.bias is None and _conv_ok(blk.conv3, 1, 1) and blk.conv3.bias is None
            and _conv_ok(sa.conv, 3) and sa.conv.bias is None and sa.conv.in_channels == C and (C // sa.conv.groups) % 8 == 0
            and blk.conv1.in_channels % 8 == 0 and (self.ds_conv is not None or blk.conv1.in_channels == blk.conv3.out_channels)
            and fc_ok(sa.fc1) and fc_ok(sa.fc2) and sa.fc1.in_channels == C and sa.fc2.out_channels == C
            and sa.fc1.out_channels % 8 == 0 and self.act0 > 0 and self.act1 > 0
            and all(_bn_static_ok(b) for b in (blk.bn1, blk.bn3, sa.bn0, sa.bn1)))
        self.params = [blk.conv1.weight, blk.bn1.weight, blk.bn1.bias, sa.conv.weight, sa.bn0.weight, sa.bn0.bias, sa.fc1.weight,
                       sa.fc1.bias, sa.bn1.weight, sa.bn1.bias, sa.fc2.weight, sa.fc2.bias, blk.conv3.weight, blk.bn3.weight,
                       blk.bn3.bias] + ([self.ds_conv.weight, self.ds_bn.weight, self.ds_bn.bias] if self.ds_conv is not None else [])


_SA_PLANS = weakref.WeakKeyDictionary()
_SASIZES = _lib.register_cache({})


def _sa_plan(blk):
    p = _SA_PLANS.get(blk)
    if p is None:
        p = _SA_PLANS[blk] = _SABlockPlan(blk)
    return p


def _sa_sizes(L, N, Cin, Cw, A, G, H, W, Cout=None, HWo=None):
    """Cout / HWo: output channels / pixels of the block (a stage-opening block changes both; default: identity shortcut)"""
    Cout, HWo = Cout or Cin, HWo or H * W
    k = (N, Cin, Cw, A, G, H, W, Cout, HWo)
    v = _SASIZES.get(k)
    if v is None:
        HW = H * W
        ws = max(int(L.cot_conv1x1_workspace(N, Cin, Cw, HW, 0)), int(L.cot_conv1x1_workspace(N, Cw, Cout, HWo, 0)),
                 int(L.cot_conv1x1_workspace(N, Cin, Cout, HWo, 0)), int(L.cot_conv3x3g_workspace(N, Cw, Cw, G, H, W)),
                 int(L.cot_conv1x1_workspace(1, Cw, A, N, 1)), int(L.cot_conv1x1_workspace(1, A, Cw, N, 1)))  # (the gate's fc layers)
        v = _SASIZES[k] = (ws, int(L.cot_bn_act_workspace(N, Cw)), int(L.cot_bn_act_workspace(N, Cout)), int(L.cot_bn_act_workspace(1, A)))
    return v


class _SplitAttnBlockNode(Function):
    @staticmethod
    @_one_stream_query
    def forward(ctx, blk, x, *params):
        L = _lib.lib()
        sp = _sa_plan(blk)
        N, Cin, H, W = x.shape
        Cw, A, G = sp.conv.out_channels, sp.fc1.out_channels, sp.conv.groups
        Cout = sp.conv3.out_channels
        Ho, Wo = (H // 2, W // 2) if sp.avd_post else (H, W)  # (even planes: sa_block_eligible)
        HW, HWo = H * W, Ho * Wo
        dev, st = x.device, _stream()
        ws_bytes, nws_w, nws_o, nws_a = _sa_sizes(L, N, Cin, Cw, A, G, H, W, Cout, HWo)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
        masks = _masks(L, H, W, dev)
        new = lambda c, h=H, w=W: torch.empty((N, c, h, w), dtype=x.dtype, device=dev)  # noqa: E731
        stat = lambda c, nws: torch.empty(2 * c + nws, dtype=torch.float32, device=dev)  # noqa: E731
        c1, a1 = new(Cw), _new_guarded(N, Cw, H, W, x.dtype, dev)  # (the 3x3 weight gradient reads a1 shifted: margins)
        _ck(L.cot_conv1x1_forward(_p(x), None, Cin, _p(sp.conv1.weight), None, _p(c1), N, Cin, Cw, HW, BF16, st), "cot_conv1x1_forward")
        s_1 = stat(Cw, nws_w)
        _bn_fwd(L, c1, a1, sp.bn1, s_1, 2 * Cw, N, Cw, HW, 1)
        c2, b2 = new(Cw), new(Cw)
        _conv3x3_fwd(L, sp.conv, a1, c2, masks, ws, N, Cw, G, H, W)
        s_0 = stat(Cw, nws_w)
        _bn_fwd(L, c2, b2, sp.bn0, s_0, 2 * Cw, N, Cw, HW, sp.act0)
        # the gate: pooled descriptor [N, Cw] -> fc1 -> BatchNorm over the batch + act -> fc2 -> x * sigmoid(logits)
        # (round 6: descriptors channel-major [.][N], as in the CoT layer's se branch -- the two fc layers are 1x1 convolutions over ONE
        # image whose N pixels are the batch, on the library's kernels; rounds 4-5 ran them as torch.addmm / matmul, i.e. vendor GEMMs:
        # 174 launches + their bias-gradient reductions per SE-CoTNetD-152 step)
        row = lambda c: torch.empty((c, N), dtype=x.dtype, device=dev)  # noqa: E731
        gap, hpre, h, logitsT = row(Cw), row(A), row(A), row(Cw)
        _ck(L.cot_radix_gap_t(_p(b2), None, _p(gap), N, Cw, HW, BF16, st), "cot_radix_gap_t")
        _ck(L.cot_conv1x1_forward(_p(gap), None, Cw, _p(sp.fc1.weight), _p(sp.fc1.bias), _p(hpre), 1, Cw, A, N, BF16, st), "cot_conv1x1_forward")
        s_s = stat(A, nws_a)
        _bn_fwd(L, hpre, h, sp.sbn, s_s, 2 * A, 1, A, N, sp.act1)  # (one image, N pixels: the statistics run over the batch)
        _ck(L.cot_conv1x1_forward(_p(h), None, A, _p(sp.fc2.weight), _p(sp.fc2.bias), _p(logitsT), 1, A, Cw, N, BF16, st), "cot_conv1x1_forward")
        logits = logitsT.t().contiguous()  # [N][Cw]: the gate kernels index it by plane n * Cw + c
        out2 = new(Cw)
        _ck(L.cot_se_gate(_p(b2), _p(logits), _p(out2), N * Cw, HW, BF16, st), "cot_se_gate")
        if sp.avd_post:  # anti-aliased down-sampling behind conv2 (cotnet_hybrid.py:196-199; blur_pool.py:53-58)
            out2p = new(Cw, Ho, Wo)
            _ck(L.cot_blurpool3x3s2_forward(_p(out2), _p(out2p), N * Cw, H, W, BF16, st), "cot_blurpool3x3s2_forward")
        else:
            out2p = out2
        c3, y = new(Cout, Ho, Wo), new(Cout, Ho, Wo)
        _ck(L.cot_conv1x1_forward(_p(out2p), None, Cw, _p(sp.conv3.weight), None, _p(c3), N, Cw, Cout, HWo, BF16, st), "cot_conv1x1_forward")
        if sp.ds_conv is not None:  # projection shortcut: bn(conv1x1([avgpool2x2](x)))
            if sp.ds_pool2:
                xs = new(Cin, Ho, Wo)
                _ck(L.cot_avgpool2x2s2_forward(_p(x), _p(xs), N * Cin, H, W, BF16, st), "cot_avgpool2x2s2_forward")
            else:
                xs = x
            d0, res = new(Cout, Ho, Wo), new(Cout, Ho, Wo)
            _ck(L.cot_conv1x1_forward(_p(xs), None, Cin, _p(sp.ds_conv.weight), None, _p(d0), N, Cin, Cout, HWo, BF16, st), "cot_conv1x1_forward")
            s_d = stat(Cout, nws_o)
            _bn_fwd(L, d0, res, sp.ds_bn, s_d, 2 * Cout, N, Cout, HWo, 0)
        else:
            xs, d0, res, s_d = None, None, x, None
        s_3 = stat(Cout, nws_o)
        ps = _drop_path_scale(blk, N, dev)
        m3 = _relu_mask(L, N, Cout, HWo, dev)
        _bn_fwd(L, c3, y, sp.bn3, s_3, 2 * Cout, N, Cout, HWo, 1, residual=res, ps=ps, mask=m3)
        ctx.blk, ctx.has_ps, ctx.has_mask = blk, ps is not None, m3 is not None
        ctx.save_for_backward(x, c1, a1, s_1, c2, b2, s_0, gap, hpre, h, s_s, logits, out2, c3, y, s_3, out2p,
                              *((d0, s_d, xs) if sp.ds_conv is not None else ()),
                              *((m3,) if m3 is not None else ()), *((ps,) if ps is not None else ()))
        return y

    @staticmethod
    @_one_stream_query
    def backward(ctx, gout):
        L = _lib.lib()
        blk = ctx.blk
        sp = _sa_plan(blk)
        t = ctx.saved_tensors
        x, c1, a1, s_1, c2, b2, s_0, gap, hpre, h, s_s, logits, out2, c3, y, s_3, out2p = t[:17]
        nds = 3 if sp.ds_conv is not None else 0
        d0, s_d, xs = t[17:20] if nds else (None, None, None)
        m3 = t[17 + nds] if ctx.has_mask else None
        ps = t[-1] if ctx.has_ps else None
        N, Cin, H, W = x.shape
        Cw, A, G = sp.conv.out_channels, sp.fc1.out_channels, sp.conv.groups
        Cout, Ho, Wo = y.shape[1], y.shape[2], y.shape[3]
        HW, HWo = H * W, Ho * Wo
        dev, st = x.device, _stream()
        ws_bytes, nws_w, nws_o, nws_a = _sa_sizes(L, N, Cin, Cw, A, G, H, W, Cout, HWo)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
        masks = _masks(L, H, W, dev)
        side = _Side(dev, ws_bytes, ws, sp.params)
        gout = gout.contiguous()
        fold = sp.ds_conv is None and m3 is not None and not sp.avd_post and clf._res_fold_ok(L, N, Cin, Cw, HW)
        g_c3, g_res = torch.empty_like(c3), (None if fold else torch.empty_like(c3))  # (fold: the residual's gradient goes into conv1's data gradient)
        d_bn3_w, d_bn3_b = _bn_bwd(L, gout, c3, y, g_c3, sp.bn3, s_3, N, Cout, HWo, 1, nws_o, dres=g_res, ps=ps, mask=m3)
        g_out2p = torch.empty_like(out2p)
        _ck(L.cot_conv1x1_backward_data(_p(g_c3), _p(sp.conv3.weight), _p(g_out2p), None, Cw, 0, _p(ws), N, Cw, Cout, HWo, BF16, st),
            "cot_conv1x1_backward_data")
        g_w3 = grad_sink.out_like(sp.conv3.weight)
        side.run(lambda st_, a_=(_p(g_c3), _p(out2p), None, Cw, _p(g_w3), None, _p(side.ws), N, Cw, Cout, HWo, BF16): _ck(L.cot_conv1x1_backward_weight(*a_, st_), "cot_conv1x1_backward_weight"), g_c3, out2p)
        if sp.avd_post:
            g_out2 = torch.empty_like(out2)
            _ck(L.cot_blurpool3x3s2_backward(_p(g_out2p), _p(g_out2), N * Cw, H, W, BF16, st), "cot_blurpool3x3s2_backward")
        else:
            g_out2 = g_out2p
        # gate: dx of x * sigmoid(l) and dl in one pass; then the two fc layers (GEMMs on [N, .] descriptors) and their BatchNorm
        g_b2, g_log = torch.empty_like(b2), torch.empty_like(logits)
        _ck(L.cot_se_gate_backward(_p(g_out2), _p(b2), _p(logits), _p(g_b2), _p(g_log), N * Cw, HW, BF16, st), "cot_se_gate_backward")
        row = lambda c: torch.empty((c, N), dtype=x.dtype, device=dev)  # noqa: E731
        g_logT = g_log.t().contiguous()  # [Cw][N]
        g_h, g_hpre, g_gapT = row(A), row(A), row(Cw)
        _ck(L.cot_conv1x1_backward_data(_p(g_logT), _p(sp.fc2.weight), _p(g_h), None, A, 0, _p(ws), 1, A, Cw, N, BF16, st), "cot_conv1x1_backward_data")
        g_fc2_w, g_fc2_b = grad_sink.out_like(sp.fc2.weight), grad_sink.out_like(sp.fc2.bias)
        side.run(lambda st_, a_=(_p(g_logT), _p(h), None, A, _p(g_fc2_w), _p(g_fc2_b), _p(side.ws), 1, A, Cw, N, BF16): _ck(L.cot_conv1x1_backward_weight(*a_, st_), "cot_conv1x1_backward_weight"), g_logT, h)
        d_sbn_w, d_sbn_b = _bn_bwd(L, g_h, hpre, None, g_hpre, sp.sbn, s_s, 1, A, N, sp.act1, nws_a)
        _ck(L.cot_conv1x1_backward_data(_p(g_hpre), _p(sp.fc1.weight), _p(g_gapT), None, Cw, 0, _p(ws), 1, Cw, A, N, BF16, st), "cot_conv1x1_backward_data")
        g_fc1_w, g_fc1_b = grad_sink.out_like(sp.fc1.weight), grad_sink.out_like(sp.fc1.bias)
        side.run(lambda st_, a_=(_p(g_hpre), _p(gap), None, Cw, _p(g_fc1_w), _p(g_fc1_b), _p(side.ws), 1, Cw, A, N, BF16): _ck(L.cot_conv1x1_backward_weight(*a_, st_), "cot_conv1x1_backward_weight"), g_hpre, gap)
        g_b2.add_((g_gapT.t().float() / HW).to(g_b2.dtype).reshape(N, Cw, 1, 1))  # d mean_hw: the same value for every pixel of a plane
        g_c2 = g_out2  # (reuse: consumed by the gate's backward)
        d_bn0_w, d_bn0_b = _bn_bwd(L, g_b2, c2, None, g_c2, sp.bn0, s_0, N, Cw, HW, sp.act0, nws_w)
        g_wc = grad_sink.out_like(sp.conv.weight)
        side.run(lambda st_, a_=(_p(g_c2), _p(a1), _p(g_wc), _p(masks), _p(side.ws), N, Cw, Cw, G, H, W, BF16, _guard_elems(a1)): _ck(L.cot_conv3x3g_backward_weight_guarded(*a_, st_), "cot_conv3x3g_backward_weight"), g_c2, a1, masks)
        g_a1 = g_b2  # (reuse: consumed by bn0's backward)
        _conv3x3_dgrad(L, sp.conv, g_c2, g_a1, 0, masks, ws, N, Cw, G, H, W)
        g_c1 = torch.empty_like(c1)
        d_bn1_w, d_bn1_b = _bn_bwd(L, g_a1, c1, None, g_c1, sp.bn1, s_1, N, Cw, HW, 1, nws_w)
        g_ds = ()
        if sp.ds_conv is not None:  # projection shortcut: BatchNorm, 1x1 convolution [, the 2 x 2 average] backwards -> first contribution to dx
            g_d0 = torch.empty_like(g_c3) if side.on else g_c3  # (g_c3 is still read by conv3's weight gradient on the side stream)
            d_ds_w, d_ds_b = _bn_bwd(L, g_res, d0, None, g_d0, sp.ds_bn, s_d, N, Cout, HWo, 0, nws_o)
            g_xs = torch.empty_like(xs)
            _ck(L.cot_conv1x1_backward_data(_p(g_d0), _p(sp.ds_conv.weight), _p(g_xs), None, Cin, 0, _p(ws), N, Cin, Cout, HWo, BF16, st),
                "cot_conv1x1_backward_data")
            if sp.ds_pool2:
                gx = torch.empty_like(x)
                _ck(L.cot_avgpool2x2s2_backward(_p(g_xs), _p(gx), N * Cin, H, W, BF16, st), "cot_avgpool2x2s2_backward")
            else:
                gx = g_xs
            g_wd = grad_sink.out_like(sp.ds_conv.weight)
            side.run(lambda st_, a_=(_p(g_d0), _p(xs), None, Cin, _p(g_wd), None, _p(side.ws), N, Cin, Cout, HWo, BF16): _ck(L.cot_conv1x1_backward_weight(*a_, st_), "cot_conv1x1_backward_weight"), g_d0, xs)
            g_ds = (g_wd, d_ds_w, d_ds_b)
        else:
            gx = g_res  # identity shortcut: the residual's gradient is the first contribution to dx
        g_w1 = grad_sink.out_like(sp.conv1.weight)
        side.run(lambda st_, a_=(_p(g_c1), _p(x), None, Cin, _p(g_w1), None, _p(side.ws), N, Cin, Cw, HW, BF16): _ck(L.cot_conv1x1_backward_weight(*a_, st_), "cot_conv1x1_backward_weight"), g_c1, x)
        if fold:
            gx = torch.empty_like(x)
            _ck(L.cot_conv1x1_backward_data_relu_res(_p(g_c1), _p(sp.conv1.weight), _p(gx), _p(gout), _p(m3), N, Cin, Cw, HW, BF16, st),
                "cot_conv1x1_backward_data_relu_res")
        else:
            _ck(L.cot_conv1x1_backward_data(_p(g_c1), _p(sp.conv1.weight), _p(gx), None, Cin, 1, _p(ws), N, Cin, Cw, HW, BF16, st),
                "cot_conv1x1_backward_data")
        side.join()
        return (None, gx, g_w1, d_bn1_w, d_bn1_b, g_wc, d_bn0_w, d_bn0_b, g_fc1_w, g_fc1_b, d_sbn_w, d_sbn_b, g_fc2_w, g_fc2_b, g_w3,
                d_bn3_w, d_bn3_b) + g_ds


def sa_block_eligible(blk, x):
    """training-mode cotnet_hybrid.CoTBottleneck with a SplitAttnConv2d(radix=1) conv2, identity shortcut, on a bf16 NCHW tensor"""
    if not (clf.ENABLED and blk.training and (x.is_cuda or not clf._DEVICE_ONLY) and x.dim() == 4 and x.dtype == torch.bfloat16
            and x.is_contiguous() and x.data_ptr() % 16 == 0):
        return False
    sp = _sa_plan(blk)
    return (sp.static_ok and x.shape[1] == sp.conv1.in_channels and sp.conv1.weight.dtype == torch.bfloat16
            and sp.conv.weight.dtype == torch.bfloat16 and sp.fc1.weight.dtype == torch.bfloat16 and sp.conv3.weight.dtype == torch.bfloat16
            and sp.bn1.weight.dtype == torch.float32 and sp.bn1.training and sp.bn0.training and sp.sbn.training and sp.bn3.training
            and x.shape[0] >= 2 and not (sp.avd_post and (x.shape[2] % 2 or x.shape[3] % 2))
            and (sp.ds_bn is None or sp.ds_bn.training))


def sa_block_forward(blk, x):
    NODE_COUNTS["split_attn_block"] += 1
    return _SplitAttnBlockNode.apply(blk, x, *_sa_plan(blk).params)

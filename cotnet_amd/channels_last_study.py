"""STUDY: the CotLayer (models/cotnet.py:79-104) forward and backward on channels-last tensors [N*H*W][C] as the
sequence of launches a single-node implementation would issue, on the study kernels `cot_study_*` (csrc/gemm_kc.hip, bn_nhwc.hip,
gn9_nhwc.hip, radix_nhwc.hip) plus the NHWC aggregation of the C ABI.  Not wired into any model: the host-emulated tests
(tests/test_kernels_emulated.py::test_cot_layer_*_composed_from_the_channels_last_study_kernels) check it against the module's
formula, scripts/bench_cot_layer_channels_last.py times it on the GPU beside the NCHW single-node layer.

`lib` is the loaded library (ctypes), `stream` a ctypes void pointer or None; tensors live wherever `x` lives.  Training mode only
(batch statistics; running statistics are not updated here), bf16 only."""
import ctypes

import torch

from . import _lib

BF = _lib.COT_BF16
_F = ctypes.c_float


def _p(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(None)


def _stream(t):
    """the launch stream for tensors like `t`: torch's current stream on the GPU, none for the host emulator's CPU tensors"""
    return ctypes.c_void_p(torch.cuda.current_stream(t.device).cuda_stream) if t.is_cuda else None


def _ok(rc, what):
    if rc != 0:
        raise RuntimeError(f"cotnet_amd channels-last study: {what} returned {rc}")


_ZEROS = {}


def _zeros(device):
    """one block of zeros per device: the source of a 3x3 tap outside the image"""
    z = _ZEROS.get(device)
    if z is None:
        z = _ZEROS[device] = torch.zeros(64, dtype=torch.bfloat16, device=device)
    return z


class Plan:
    """repacked / transposed copies of a CotLayer's weights (bf16): what a channels-last node would keep per optimizer step"""

    def __init__(self, layer):
        def w2(c):
            return c.weight.detach().reshape(c.out_channels, -1).contiguous()
        self.layer = layer
        D = layer.dim
        self.D, self.G = D, D // 8
        ke = layer.key_embed[0]
        self.groups = ke.groups
        Kc = D // self.groups
        self.wr = ke.weight.detach().permute(0, 2, 3, 1).contiguous()                                   # [Co][3][3][Kc]
        self.wr_t = (ke.weight.detach().view(self.groups, Kc, Kc, 3, 3).flip(3, 4).permute(0, 2, 3, 4, 1)
                     .reshape(D, 3, 3, Kc).contiguous())                                                   # data gradient's repack
        self.w_e0, self.w_e3, self.w_v = w2(layer.embed[0]), w2(layer.embed[3]), w2(layer.conv1x1[0])
        self.w_s0, self.w_s3 = w2(layer.se[0]), w2(layer.se[3])
        self.zeros = _zeros(ke.weight.device)


class _Ops:
    def __init__(self, lib, stream, like):
        self.L, self.s, self.like = lib, stream, like
        lib.cot_study_conv1x1_nhwc_wgrad_workspace.restype = ctypes.c_size_t
        lib.cot_study_conv3x3g_nhwc_wgrad_workspace.restype = ctypes.c_size_t

    def new(self, *shape, dtype=torch.bfloat16):
        return torch.empty(shape, dtype=dtype, device=self.like.device)

    def gemm(self, x1, x2, k1, w, b, rows, y=None, acc=0):
        y = self.new(rows, w.shape[0]) if y is None else y
        _ok(self.L.cot_study_conv1x1_nhwc(_p(x1), _p(x2), k1, _p(w), _p(b), _p(y), acc, rows, w.shape[0], w.shape[1], 0, self.s), "conv1x1")
        return y

    def dgrad(self, dy, w, rows, y=None, acc=0, col0=0, ncols=None):
        """y[:, :ncols] (+)= dy @ w[:, col0 : col0 + ncols] straight from the convolution's weight w [Co][Ci] (no transposed copy)"""
        K, ldb = w.shape
        ncols = ldb if ncols is None else ncols
        y = self.new(rows, ncols) if y is None else y
        wp = ctypes.c_void_p(w.data_ptr() + 2 * col0)
        _ok(self.L.cot_study_conv1x1_nhwc_dgrad(_p(dy), wp, _p(y), acc, rows, ncols, K, ldb, y.shape[1], 0, self.s), "conv1x1 dgrad")
        return y

    def wgrad(self, xin, dy, rows, dw=None, col0=0):
        """dW [Co][Ci] = dy^T xin -- or, with `dw` given, into the column window [col0, col0 + Ci) of that wider gradient"""
        Ci, Co = xin.shape[1], dy.shape[1]
        ws = self.new(self.L.cot_study_conv1x1_nhwc_wgrad_workspace(rows, Ci, Co, 0), dtype=torch.uint8)
        if dw is None:
            dw = self.new(Co, Ci)
        _ok(self.L.cot_study_conv1x1_nhwc_wgrad_window(_p(xin), _p(dy), ctypes.c_void_p(dw.data_ptr() + 2 * col0), dw.shape[1], _p(ws), rows, Ci,
                                                       Co, 0, self.s), "conv1x1 wgrad")
        return dw

    def colsum(self, t, rows):
        o = self.new(t.shape[1])
        ws = self.new(self.L.cot_study_nhwc_col_sum_workspace(rows, t.shape[1]), dtype=torch.float32)
        _ok(self.L.cot_study_nhwc_col_sum(_p(t), _p(o), _p(ws), rows, t.shape[1], BF, self.s), "col_sum")
        return o

    def bn_f(self, t, mod, act, rows):
        C = t.shape[-1]
        y, mean, rstd = torch.empty_like(t), self.new(C, dtype=torch.float32), self.new(C, dtype=torch.float32)
        ws = self.new(self.L.cot_study_bn_nhwc_workspace(rows, C, BF), dtype=torch.float32)
        _ok(self.L.cot_study_bn_nhwc_forward(_p(t), _p(None), _p(y), _p(mod.weight.detach()), _p(mod.bias.detach()), _p(mean), _p(rstd),
                                             _p(None), _p(None), _p(None), _p(ws), rows, C, _F(mod.eps), _F(0.1), act, BF, self.s), "bn fwd")
        return y, (mean, rstd)

    def bn_b(self, dy, t, y, st, mod, act, rows):
        C = t.shape[-1]
        dx, dg, db = torch.empty_like(t), self.new(C, dtype=torch.float32), self.new(C, dtype=torch.float32)
        ws = self.new(self.L.cot_study_bn_nhwc_workspace(rows, C, BF), dtype=torch.float32)
        _ok(self.L.cot_study_bn_nhwc_backward(_p(dy), _p(t), _p(y), _p(dx), _p(None), _p(mod.weight.detach()), _p(mod.bias.detach()), _p(st[0]),
                                              _p(st[1]), _p(dg), _p(db), _p(ws), rows, C, act, BF, self.s), "bn bwd")
        return dx, dg, db


def forward(lib, plan, x, N, H, W, stream=None):
    """x [N][H][W][D] bf16 contiguous -> (out [N*H*W][D], saved tensors for `backward`)"""
    ly, D, G = plan.layer, plan.D, plan.G
    HW, M = H * W, N * H * W
    o = _Ops(lib, stream, x)
    xm = x.reshape(M, D)
    k_pre = o.new(M, D)
    _ok(lib.cot_study_conv3x3g_nhwc(_p(x), _p(plan.wr), _p(plan.zeros), _p(k_pre), 0, N, H, W, D, D, plan.groups, stream), "conv3x3")
    k, k_st = o.bn_f(k_pre, ly.key_embed[1], 1, M)
    e0_pre = o.gemm(xm, k, D, plan.w_e0, None, M)                      # embed[0] on [x | k], no cat
    e0, e0_st = o.bn_f(e0_pre, ly.embed[1], 1, M)
    e3 = o.gemm(e0, None, e0.shape[1], plan.w_e3, ly.embed[3].bias.detach(), M)
    gn = ly.embed[4]
    wn, gm, gr = torch.empty_like(e3), o.new(N * G, dtype=torch.float32), o.new(N * G, dtype=torch.float32)
    _ok(lib.cot_study_group_norm9_nhwc_forward(_p(e3), _p(gn.weight.detach()), _p(gn.bias.detach()), _p(wn), _p(gm), _p(gr), N, 9 * G, HW,
                                               _F(gn.eps), BF, stream), "gn9")
    v_pre = o.gemm(xm, None, D, plan.w_v, None, M)
    v, v_st = o.bn_f(v_pre, ly.conv1x1[1], 0, M)
    geo = _lib.AggGeom(N, D, H, W, 1, G, 3, 3, 1, 1, 1, 1, 1, 1)
    agg = o.new(M, D)
    _ok(lib.cot_agg_forward(_p(v), _p(wn), _p(agg), ctypes.byref(geo), BF, _lib.COT_NHWC, stream), "aggregation")
    y, y_st = o.bn_f(agg, ly.bn, 2, M)
    gap = o.new(N, D)
    _ok(lib.cot_study_radix_nhwc_gap(_p(y), _p(k), _p(gap), N, HW, D, BF, stream), "radix gap")
    s0_pre = o.gemm(gap, None, D, plan.w_s0, ly.se[0].bias.detach(), N)   # the `se` branch: GEMMs / BatchNorm on the [N][C] descriptor
    s0, s0_st = o.bn_f(s0_pre, ly.se[1], 1, N)
    logits = o.gemm(s0, None, s0.shape[1], plan.w_s3, ly.se[3].bias.detach(), N)
    attn = o.new(N, D, 2)                                                # softmax over the radix pair of each (image, channel)
    _ok(lib.cot_study_radix_softmax2(_p(logits), _p(attn), ctypes.c_int64(N * D), BF, stream), "radix softmax")
    out = o.new(M, D)
    _ok(lib.cot_study_radix_nhwc_mix(_p(y), _p(k), _p(attn), _p(out), N, HW, D, BF, stream), "radix mix")
    saved = dict(x=x, xm=xm, k_pre=k_pre, k=k, k_st=k_st, e0_pre=e0_pre, e0=e0, e0_st=e0_st, e3=e3, wn=wn, gm=gm, gr=gr, v_pre=v_pre, v=v,
                 v_st=v_st, agg=agg, y=y, y_st=y_st, gap=gap, s0_pre=s0_pre, s0=s0, s0_st=s0_st, attn=attn, geo=geo, N=N, H=H, W=W)
    return out, saved


def backward(lib, plan, sv, gout, stream=None):
    """gout [N*H*W][D] -> (gx [N*H*W][D], {parameter name: gradient})"""
    ly, D, G = plan.layer, plan.D, plan.G
    N, H, W = sv["N"], sv["H"], sv["W"]
    HW, M = H * W, N * H * W
    o = _Ops(lib, stream, gout)
    g = {}
    gattn = o.new(N, D, 2)
    _ok(lib.cot_study_radix_nhwc_mix_backward_reduce(_p(gout), _p(sv["y"]), _p(sv["k"]), _p(gattn), N, HW, D, BF, stream), "radix reduce")
    glog = o.new(N, 2 * D)
    _ok(lib.cot_study_radix_softmax2_backward(_p(sv["attn"]), _p(gattn), _p(glog), ctypes.c_int64(N * D), BF, stream), "radix softmax backward")
    g["se.3.weight"], g["se.3.bias"] = o.wgrad(sv["s0"], glog, N), o.colsum(glog, N)
    gs0 = o.dgrad(glog, plan.w_s3, N)
    gs0_pre, g["se.1.weight"], g["se.1.bias"] = o.bn_b(gs0, sv["s0_pre"], sv["s0"], sv["s0_st"], ly.se[1], 1, N)
    g["se.0.weight"], g["se.0.bias"] = o.wgrad(sv["gap"], gs0_pre, N), o.colsum(gs0_pre, N)
    ggap = o.dgrad(gs0_pre, plan.w_s0, N)
    gy, gk = o.new(M, D), o.new(M, D)
    _ok(lib.cot_study_radix_nhwc_mix_backward_apply(_p(gout), _p(sv["attn"]), _p(ggap), _p(gy), _p(gk), N, HW, D, BF, stream), "radix apply")
    gagg, g["bn.weight"], g["bn.bias"] = o.bn_b(gy, sv["agg"], None, sv["y_st"], ly.bn, 2, M)
    gv, gwn = o.new(M, D), torch.empty_like(sv["wn"])
    _ok(lib.cot_agg_backward(_p(gagg), _p(sv["v"]), _p(sv["wn"]), _p(gv), _p(gwn), ctypes.byref(sv["geo"]), BF, _lib.COT_NHWC, stream),
        "aggregation backward")
    gv_pre, g["conv1x1.1.weight"], g["conv1x1.1.bias"] = o.bn_b(gv, sv["v_pre"], None, sv["v_st"], ly.conv1x1[1], 0, M)
    g["conv1x1.0.weight"] = o.wgrad(sv["xm"], gv_pre, M)
    gx = o.dgrad(gv_pre, plan.w_v, M)                                                          # the values' branch starts gx
    gn = ly.embed[4]
    ge3, dgg, dgb = torch.empty_like(sv["e3"]), torch.empty_like(gn.weight.detach()), torch.empty_like(gn.bias.detach())
    gws = o.new(N * 9 * G * 2, dtype=torch.float32)
    _ok(lib.cot_study_group_norm9_nhwc_backward(_p(gwn), _p(sv["e3"]), _p(sv["gm"]), _p(sv["gr"]), _p(gn.weight.detach()), _p(ge3), _p(dgg),
                                                _p(dgb), _p(gws), N, 9 * G, HW, BF, stream), "gn9 backward")
    g["embed.4.weight"], g["embed.4.bias"] = dgg, dgb
    g["embed.3.weight"], g["embed.3.bias"] = o.wgrad(sv["e0"], ge3, M), o.colsum(ge3, M)
    ge0 = o.dgrad(ge3, plan.w_e3, M)
    ge0_pre, g["embed.1.weight"], g["embed.1.bias"] = o.bn_b(ge0, sv["e0_pre"], sv["e0"], sv["e0_st"], ly.embed[1], 1, M)
    gw0 = o.new(ge0_pre.shape[1], 2 * D)                                                      # embed[0]'s weight gradient: the [x | k] slabs
    o.wgrad(sv["xm"], ge0_pre, M, dw=gw0, col0=0)                                             # ... written into their column windows
    o.wgrad(sv["k"], ge0_pre, M, dw=gw0, col0=D)
    g["embed.0.weight"] = gw0
    o.dgrad(ge0_pre, plan.w_e0, M, y=gx, acc=1, col0=0, ncols=D)                               # gx += embed[0]'s [x | .] half
    o.dgrad(ge0_pre, plan.w_e0, M, y=gk, acc=1, col0=D, ncols=D)                               # gk += its [. | k] half
    gk_pre, g["key_embed.1.weight"], g["key_embed.1.bias"] = o.bn_b(gk, sv["k_pre"], sv["k"], sv["k_st"], ly.key_embed[1], 1, M)
    Kc = D // plan.groups
    ws3 = o.new(lib.cot_study_conv3x3g_nhwc_wgrad_workspace(N, H, W, D, D, plan.groups, 0), dtype=torch.uint8)
    dwr = o.new(D, 9, Kc)
    _ok(lib.cot_study_conv3x3g_nhwc_wgrad(_p(sv["x"]), _p(gk_pre), _p(plan.zeros), _p(dwr), _p(ws3), N, H, W, D, D, plan.groups, 0, stream),
        "conv3x3 wgrad")
    g["key_embed.0.weight"] = dwr.view(D, 3, 3, Kc).permute(0, 3, 1, 2)
    _ok(lib.cot_study_conv3x3g_nhwc(_p(gk_pre), _p(plan.wr_t), _p(plan.zeros), _p(gx), 1, N, H, W, D, D, plan.groups, stream), "conv3x3 dgrad")
    return gx, g


# ---- the stride-1 Bottleneck without a projection (models/cotnet.py:228-264; five of layer3's six blocks, two of layer4's three) ----------
class BlockPlan:
    def __init__(self, blk):
        assert blk.downsample is None and blk.avd is None and blk.se is None and blk.drop_block is None and blk.drop_path is None
        self.blk = blk
        self.layer = Plan(blk.conv2)
        self.w1 = blk.conv1.weight.detach().reshape(blk.conv1.out_channels, -1).contiguous()
        self.w3 = blk.conv3.weight.detach().reshape(blk.conv3.out_channels, -1).contiguous()


def block_forward(lib, plan, x, N, H, W, stream=None):
    """x [N][H][W][C] -> (out [N*H*W][C], saved)"""
    blk, M = plan.blk, N * H * W
    o = _Ops(lib, stream, x)
    xm = x.reshape(M, x.shape[-1])
    t1_pre = o.gemm(xm, None, xm.shape[1], plan.w1, None, M)
    t1, t1_st = o.bn_f(t1_pre, blk.bn1, 1, M)
    t2, sv = forward(lib, plan.layer, t1.view(N, H, W, -1), N, H, W, stream)
    t3_pre = o.gemm(t2, None, t2.shape[1], plan.w3, None, M)
    C = t3_pre.shape[1]
    out, mean, rstd = torch.empty_like(t3_pre), o.new(C, dtype=torch.float32), o.new(C, dtype=torch.float32)
    ws = o.new(lib.cot_study_bn_nhwc_workspace(M, C, BF), dtype=torch.float32)
    _ok(lib.cot_study_bn_nhwc_forward(_p(t3_pre), _p(xm), _p(out), _p(blk.bn3.weight.detach()), _p(blk.bn3.bias.detach()), _p(mean), _p(rstd),
                                      _p(None), _p(None), _p(None), _p(ws), M, C, _F(blk.bn3.eps), _F(0.1), 1, BF, stream), "bn3 + residual")
    return out, dict(xm=xm, t1_pre=t1_pre, t1=t1, t1_st=t1_st, layer=sv, t2=t2, t3_pre=t3_pre, out=out, st3=(mean, rstd), N=N, H=H, W=W)


def block_backward(lib, plan, sv, gout, stream=None):
    """-> (gx [N*H*W][C], {parameter name: gradient})"""
    blk = plan.blk
    N, H, W = sv["N"], sv["H"], sv["W"]
    M = N * H * W
    o = _Ops(lib, stream, gout)
    g = {}
    C = gout.shape[1]
    g3, gx, dg, db = torch.empty_like(gout), torch.empty_like(gout), o.new(C, dtype=torch.float32), o.new(C, dtype=torch.float32)
    ws = o.new(lib.cot_study_bn_nhwc_workspace(M, C, BF), dtype=torch.float32)
    _ok(lib.cot_study_bn_nhwc_backward(_p(gout), _p(sv["t3_pre"]), _p(sv["out"]), _p(g3), _p(gx), _p(blk.bn3.weight.detach()),
                                       _p(blk.bn3.bias.detach()), _p(sv["st3"][0]), _p(sv["st3"][1]), _p(dg), _p(db), _p(ws), M, C, 1, BF, stream),
        "bn3 backward")                                                         # gx = the shortcut's share (dresidual)
    g["bn3.weight"], g["bn3.bias"] = dg, db
    g["conv3.weight"] = o.wgrad(sv["t2"], g3, M)
    g2 = o.dgrad(g3, plan.w3, M)
    g1, lg = backward(lib, plan.layer, sv["layer"], g2, stream)
    g.update({"conv2." + k: v for k, v in lg.items()})
    g1_pre, g["bn1.weight"], g["bn1.bias"] = o.bn_b(g1, sv["t1_pre"], sv["t1"], sv["t1_st"], blk.bn1, 1, M)
    g["conv1.weight"] = o.wgrad(sv["xm"], g1_pre, M)
    o.dgrad(g1_pre, plan.w1, M, y=gx, acc=1)                                    # gx += the branch
    return gx, g


class BottleneckCL(torch.autograd.Function):
    """out = BottleneckCL.apply(lib, plan, x_channels_last, *parameters): the block as ONE autograd node on channels-last tensors; the
    parameters (in `plan.blk.named_parameters()` order) are arguments only so that autograd routes their gradients"""

    @staticmethod
    def forward(ctx, lib, plan, x, *params):
        N, H, W, _ = x.shape
        out, sv = block_forward(lib, plan, x.contiguous(), N, H, W, _stream(x))
        ctx.lib, ctx.plan, ctx.sv = lib, plan, sv
        return out.view(N, H, W, -1)

    @staticmethod
    def backward(ctx, gout):
        sv = ctx.sv
        M = sv["N"] * sv["H"] * sv["W"]
        gx, g = block_backward(ctx.lib, ctx.plan, sv, gout.contiguous().view(M, -1), _stream(gout))
        names = [n for n, _ in ctx.plan.blk.named_parameters()]
        return (None, None, gx.view(gout.shape)) + tuple(g[n].reshape(p.shape).to(p.dtype) for n, p in zip(names, ctx.plan.blk.parameters()))


# ---- routing a stage (nn.Sequential of Bottlenecks, NCHW in / NCHW out) through the channels-last node where it applies ------------------
class _ToCL(torch.autograd.Function):
    """[N][C][H][W] -> [N][H][W][C] (csrc/layout_nhwc.hip); the gradient takes the way back"""

    @staticmethod
    def forward(ctx, lib, x):
        N, C, H, W = x.shape
        x = x.contiguous()  # (kept in a name: the pointer handed to the library must outlive the call)
        y = torch.empty((N, H, W, C), dtype=x.dtype, device=x.device)
        _ok(lib.cot_study_nchw_to_nhwc(_p(x), _p(y), N, C, H * W, _stream(x)), "nchw_to_nhwc")
        ctx.lib = lib
        return y

    @staticmethod
    def backward(ctx, g):
        N, H, W, C = g.shape
        g = g.contiguous()
        gx = torch.empty((N, C, H, W), dtype=g.dtype, device=g.device)
        _ok(ctx.lib.cot_study_nhwc_to_nchw(_p(g), _p(gx), N, C, H * W, _stream(g)), "nhwc_to_nchw")
        return None, gx


class _FromCL(torch.autograd.Function):
    @staticmethod
    def forward(ctx, lib, x):
        N, H, W, C = x.shape
        x = x.contiguous()
        y = torch.empty((N, C, H, W), dtype=x.dtype, device=x.device)
        _ok(lib.cot_study_nhwc_to_nchw(_p(x), _p(y), N, C, H * W, _stream(x)), "nhwc_to_nchw")
        ctx.lib = lib
        return y

    @staticmethod
    def backward(ctx, g):
        N, C, H, W = g.shape
        g = g.contiguous()
        gx = torch.empty((N, H, W, C), dtype=g.dtype, device=g.device)
        _ok(ctx.lib.cot_study_nchw_to_nhwc(_p(g), _p(gx), N, C, H * W, _stream(g)), "nchw_to_nhwc")
        return None, gx


def block_eligible(blk, x):
    """the stride-1 CoT Bottleneck in training mode, bf16, at widths the study kernels cover (CoTNet-50's layer3 / layer4: 256 / 512)"""
    from .cotnet import Bottleneck, CotLayer
    if not (isinstance(blk, Bottleneck) and isinstance(blk.conv2, CotLayer) and blk.training and torch.is_grad_enabled()):
        return False
    if blk.downsample is not None or blk.avd is not None or blk.se is not None or blk.drop_block is not None or blk.drop_path is not None:
        return False
    D = blk.conv2.dim
    return (x.dtype == torch.bfloat16 and blk.conv1.weight.dtype == torch.bfloat16 and blk.conv1.in_channels % 32 == 0 and D % 256 == 0
            and D <= 2048 and blk.conv2.key_embed[0].groups == 4 and isinstance(blk.act1, torch.nn.ReLU) and isinstance(blk.act3, torch.nn.ReLU))


def run_stage(lib, stage, x):
    """stage(x) with every run of consecutive eligible blocks on BottleneckCL between two layout changes; the other blocks as they are"""
    cl = False
    for blk in stage:
        if block_eligible(blk, x if not cl else x.permute(0, 3, 1, 2)):
            if not cl:
                x, cl = _ToCL.apply(lib, x), True
            x = BottleneckCL.apply(lib, BlockPlan(blk), x, *blk.parameters())
        else:
            if cl:
                x, cl = _FromCL.apply(lib, x), False
            x = blk(x)
    return _FromCL.apply(lib, x) if cl else x

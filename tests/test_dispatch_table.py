"""The dispatch table: which kernel variant every layer of the BASELINE configurations takes.

The library picks its kernels by shape through a stack of rules (tile shapes, image groups, split counts, kernel generations
behind 26 tuning keys).  Every rule is a place for a shape to fall onto a variant nobody measured or tested, so the table
is pinned: the models of BASELINE.json's configs 2-5 are walked on the `meta` device (shapes only), every convolution /
BatchNorm / GroupNorm / aggregation / pooling call they would make is issued against the REAL library in dry-run mode
(cot_set_tuning(26, 1): no launch, no HIP call -- works in the GPU-less container), and the recorded launches are compared
with tests/golden/dispatch_table.json.  A deliberate change of a rule = regenerate the fixture and look at the diff:

    python tests/test_dispatch_table.py --write
"""
import ctypes
import json
import os
import re
import sys

import pytest
import torch
from torch import nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
FIXTURE = os.path.join(ROOT, "tests", "golden", "dispatch_table.json")
CONFIGS = {  # BASELINE.json configs 2-5: model, per-GPU batch, image size
    "cotnet50_b80_224": ("cotnet50", 80, 224),
    "cotnext101_2x48d_b64_224": ("cotnext101_2x48d", 64, 224),
    "se_cotnetd_152_L_b64_320": ("se_cotnetd_152_L", 64, 320),
}
PTR = 0x100000  # never dereferenced in dry-run mode (16-byte aligned, non-NULL)


def _shapes(model_name, batch, size):
    """{(kind, shape tuple)} of every library-relevant call of one training forward, from a walk on the meta device"""
    import cotnet_amd
    import cotnet_amd.aggregation_zeropad as az
    from cotnet_amd import conv1x1 as c1, conv3x3g as c3, cot_layer_fused as clf, fused_bn, group_norm9 as g9, head_fused as hf, \
        pool3x3 as p3, radix_tail, stem7x7 as s7
    saved = [(m, a, getattr(m, a)) for m, a in ((c1, "MODE"), (c3, "MODE"), (g9, "MODE"), (p3, "MODE"), (hf, "MODE"), (s7, "MODE"),
                                               (clf, "ENABLED"), (fused_bn, "ENABLED"), (radix_tail, "ENABLED"))]
    old_agg = az.aggregation_zeropad
    calls = set()
    try:
        for m, a, v in saved:
            setattr(m, a, False if isinstance(v, bool) else "")  # plain modules: the walk only needs their shapes
        with torch.device("meta"):
            model = cotnet_amd.create_model(model_name, num_classes=1000).train()

        def agg(input, weight, kernel_size=3, stride=1, padding=0, dilation=1):
            calls.add(("agg", (input.shape[0], input.shape[1], input.shape[2], input.shape[3], weight.shape[2])))
            return input.new_empty(input.shape)
        az.aggregation_zeropad = agg
        for mod in model.modules():
            if hasattr(mod, "local_conv"):  # (LocalConvolution modules hold a reference to the op taken at import time)
                mod.local_conv.forward = (lambda i, w, _a=agg: _a(i, w))

        def hook(mod, inp, out):
            x = inp[0]
            if isinstance(mod, nn.Conv2d):
                N, Ci, H, W = x.shape
                k, g, s = mod.kernel_size[0], mod.groups, mod.stride[0]
                calls.add((f"conv{k}x{k}", (N, Ci, mod.out_channels, g, H, W, s, int(mod.bias is not None))))
            elif isinstance(mod, nn.BatchNorm2d):
                calls.add(("bn", (x.shape[0], x.shape[1], x.shape[2] * x.shape[3])))
            elif isinstance(mod, nn.GroupNorm):
                calls.add(("gn", (x.shape[0], x.shape[1], x.shape[2] * x.shape[3], mod.num_groups)))
            elif isinstance(mod, (nn.AvgPool2d, nn.MaxPool2d)):
                calls.add((type(mod).__name__, (x.shape[0], x.shape[1], x.shape[2], x.shape[3])))
        hooks = [m.register_forward_hook(hook) for m in model.modules()]
        model(torch.empty(batch, 3, size, size, device="meta"))
        for h in hooks:
            h.remove()
    finally:
        az.aggregation_zeropad = old_agg
        for m, a, v in saved:
            setattr(m, a, v)
    return calls


def _short(line):
    """'(kernel<..>) | launcher [T = ..] | grid G x Y | block B | lds L' -> 'kernel[args] grid=GxY block=B'"""
    kern, where, grid, block, _lds = [t.strip() for t in line.split("|")]
    name = re.sub(r"^\(|\)$", "", kern)
    name = re.sub(r"<.*", "", name)
    m = re.search(r"\[(.*)\]", where)
    args = re.sub(r"\b(T|PT|GT) = ", r"\1=", m.group(1)) if m else ""
    args = args.replace("cot::", "").replace("__bf16", "bf16").replace(" = ", "=")
    g = re.sub(r"grid (\d+) x (\d+)", r"\1x\2", grid)
    return f"{name}[{args}] grid={g} block={block.split()[-1]}"


def build_table():
    from cotnet_amd import _lib
    L = _lib.lib()
    BF = _lib.COT_BF16
    buf = ctypes.create_string_buffer(1 << 16)
    table = {}
    assert L.cot_set_tuning(26, 1) == 0
    try:
        def rec(key, rc):
            n = L.cot_launch_log(buf, len(buf))
            lines = [ln for ln in buf.value.decode().splitlines() if ln]
            table[key] = [_short(ln) for ln in lines] if rc == 0 else [f"rc={rc}: {L.cot_last_error().decode()}"]
            assert n < len(buf) - 1
        for cfg, (model, batch, size) in CONFIGS.items():
            for kind, shp in sorted(_shapes(model, batch, size)):
                tag = f"{cfg} {kind} {'x'.join(map(str, shp))}"
                if kind == "conv1x1":
                    N, Ci, Co, g, H, W, s, bias = shp
                    if s != 1:
                        H, W = (H - 1) // s + 1, (W - 1) // s + 1  # (strided projection = the stride-1 kernel on every s-th pixel)
                    HW = H * W
                    if g == 1:
                        rec(tag + " fwd", L.cot_conv1x1_forward(PTR, None, Ci, PTR, PTR if bias else None, PTR, N, Ci, Co, HW, BF, None))
                        rec(tag + " dgrad", L.cot_conv1x1_backward_data(PTR, PTR, PTR, None, Ci, 0, PTR, N, Ci, Co, HW, BF, None))
                        rec(tag + " wgrad", L.cot_conv1x1_backward_weight(PTR, PTR, None, Ci, PTR, PTR if bias else None, PTR, N, Ci, Co,
                                                                          HW, BF, None))
                        if Ci % 2 == 0 and (Ci // 2) % 32 == 0 and not bias:  # embed[0] reads [x | k] as two slabs
                            rec(tag + " fwd(two slabs)", L.cot_conv1x1_forward(PTR, PTR, Ci // 2, PTR, None, PTR, N, Ci, Co, HW, BF, None))
                    else:
                        rec(tag + " fwd", L.cot_conv1x1g_forward(PTR, PTR, PTR if bias else None, PTR, N, Ci, Co, g, HW, BF, None))
                        rec(tag + " dgrad", L.cot_conv1x1g_backward_data(PTR, PTR, PTR, 0, N, Ci, Co, g, HW, BF, None))
                        rec(tag + " wgrad", L.cot_conv1x1g_backward_weight(PTR, PTR, PTR, PTR if bias else None, PTR, N, Ci, Co, g, HW, BF,
                                                                           None))
                elif kind == "conv3x3":
                    N, Ci, Co, g, H, W, s, bias = shp
                    if s == 2 and Ci == 3 and g == 1 and not bias:  # a deep stem's first convolution: csrc/stem3x3.hip
                        rec(tag + " fwd", L.cot_stem3x3s2_forward(PTR, PTR, PTR, N, H, W, Co, BF, None))
                        rec(tag + " wgrad", L.cot_stem3x3s2_backward_weight(PTR, PTR, PTR, PTR, N, H, W, Co, BF, None))
                        continue
                    if s != 1 or bias:
                        table[tag] = ["module (MIOpen): strided / biased 3x3 convolutions are outside the library"]
                        continue
                    # (groups of 12 channels -- CoXtLayer(96).key_embed: the nodes launch pairs of groups as one group of 24 with a
                    # block-diagonal weight, cot_layer_fused._conv3x3_fwd / _dgrad)
                    from cotnet_amd import cot_layer_fused as clf
                    gl = clf._conv3x3_ws_groups(Ci, g) if Ci == Co else g
                    rec(tag + " fwd", L.cot_conv3x3g_forward(PTR, PTR, PTR, PTR, PTR, N, Ci, Co, gl, H, W, BF, None))
                    rec(tag + " dgrad", L.cot_conv3x3g_backward_data(PTR, PTR, PTR, 0, PTR, PTR, N, Ci, Co, gl, H, W, BF, None))
                    rec(tag + " wgrad", L.cot_conv3x3g_backward_weight(PTR, PTR, PTR, PTR, PTR, N, Ci, Co, g, H, W, BF, None))
                    # (the single-node Bottleneck allocates the CoT layer's input with margins: cot_layer_fused._new_guarded)
                    rec(tag + " wgrad(guarded)", L.cot_conv3x3g_backward_weight_guarded(PTR, PTR, PTR, PTR, PTR, N, Ci, Co, g, H, W, BF,
                                                                                        (W + 8) // 8 * 8, None))
                elif kind == "bn":
                    N, C, HW = shp
                    rec(tag + " fwd", L.cot_bn_act_forward(PTR, None, PTR, PTR, PTR, PTR, PTR, PTR, PTR, PTR, PTR, N, C, HW, 1e-5, 0.1, 1,
                                                           BF, None))
                    rec(tag + " bwd", L.cot_bn_act_backward(PTR, PTR, None, PTR, None, PTR, PTR, PTR, PTR, PTR, PTR, PTR, N, C, HW, 1, BF,
                                                            None))
                elif kind == "gn":
                    N, C, HW, G = shp
                    if G * 9 == C:
                        rec(tag + " fwd", L.cot_group_norm9_forward(PTR, PTR, PTR, PTR, PTR, PTR, N, C, HW, 1e-5, BF, None))
                        rec(tag + " bwd", L.cot_group_norm9_backward(PTR, PTR, PTR, PTR, PTR, PTR, PTR, PTR, PTR, N, C, HW, BF, None))
                elif kind == "agg":
                    N, C, H, W, wC = shp
                    g = _lib.AggGeom(N, C, H, W, 1, wC, 3, 3, 1, 1, 1, 1, 1, 1)
                    rec(tag + " fwd", L.cot_agg_forward(PTR, PTR, PTR, ctypes.byref(g), BF, 0, None))
                    rec(tag + " bwd", L.cot_agg_backward(PTR, PTR, PTR, PTR, PTR, ctypes.byref(g), BF, 0, None))
                elif kind in ("AvgPool2d", "MaxPool2d"):
                    N, C, H, W = shp
                    fn = L.cot_avgpool3x3s2_forward if kind == "AvgPool2d" else L.cot_maxpool3x3s2_forward
                    rec(tag + " fwd", fn(PTR, PTR, N * C, H, W, BF, None))
                # ---- the channel-major forms of the deep stages' identity Bottlenecks (cot_layer_fused._BottleneckCMNode):
                # the same 1x1 / BatchNorm entry points on channel rows (N = 1, HW' = N*HW) and the per-tensor-layout BatchNorm
                if cfg == "cotnet50_b80_224" and kind == "conv1x1" and shp[3] == 1 and shp[6] == 1 and shp[4] * shp[5] in (196, 49):
                    N, Ci, Co, g, H, W, s, bias = shp
                    M = N * H * W
                    cmt = f"{cfg} conv1x1cm {'x'.join(map(str, shp))}"
                    rec(cmt + " fwd", L.cot_conv1x1_forward(PTR, None, Ci, PTR, PTR if bias else None, PTR, 1, Ci, Co, M, BF, None))
                    rec(cmt + " dgrad", L.cot_conv1x1_backward_data(PTR, PTR, PTR, None, Ci, 0, PTR, 1, Ci, Co, M, BF, None))
                    rec(cmt + " wgrad", L.cot_conv1x1_backward_weight(PTR, PTR, None, Ci, PTR, PTR if bias else None, PTR, 1, Ci, Co, M, BF,
                                                                      None))
                    if Ci % 2 == 0 and (Ci // 2) % 32 == 0 and not bias:
                        rec(cmt + " fwd(two slabs)", L.cot_conv1x1_forward(PTR, PTR, Ci // 2, PTR, None, PTR, 1, Ci, Co, M, BF, None))
                if cfg == "cotnet50_b80_224" and kind == "bn" and shp[2] in (196, 49):
                    N, C, HW = shp
                    cmt = f"{cfg} bncm {'x'.join(map(str, shp))}"
                    rec(cmt + " fwd", L.cot_bn_act_forward(PTR, None, PTR, PTR, PTR, PTR, PTR, PTR, PTR, PTR, PTR, 1, C, N * HW, 1e-5, 0.1, 1,
                                                           BF, None))
                    rec(cmt + " bwd", L.cot_bn_act_backward(PTR, PTR, None, PTR, None, PTR, PTR, PTR, PTR, PTR, PTR, PTR, 1, C, N * HW, 1, BF,
                                                            None))
                    lyt = f"{cfg} bnlay {'x'.join(map(str, shp))}"
                    rec(lyt + " fwd", L.cot_bn_act_forward_lay(PTR, None, PTR, PTR, PTR, PTR, PTR, PTR, PTR, PTR, PTR, None, N, C, HW, 1e-5, 0.1,
                                                               1, 1 | 8, BF, None))
                    rec(lyt + " bwd", L.cot_bn_act_backward_lay(PTR, PTR, PTR, None, PTR, None, PTR, PTR, PTR, PTR, PTR, PTR, None, N, C, HW, 1,
                                                                1 | 4 | 16, BF, None))
    finally:
        L.cot_launch_log(buf, len(buf))
        assert L.cot_set_tuning(26, 0) == 0
    return table


def test_dispatch_table_is_the_pinned_one():
    table = build_table()
    want = json.load(open(FIXTURE))
    missing, extra = sorted(set(want) - set(table)), sorted(set(table) - set(want))
    assert not missing and not extra, (missing[:5], extra[:5])
    diff = {k: (want[k], table[k]) for k in want if want[k] != table[k]}
    assert not diff, f"{len(diff)} layer(s) changed their kernel -- regenerate with `python tests/test_dispatch_table.py --write` " \
                     f"after checking: {list(diff.items())[:3]}"


def test_no_benchmark_layer_falls_off_the_tuned_kernels():
    """structure of the table, independent of the exact variants: on CoTNet-50 (config 2/3) every 1x1 convolution runs on the
    third-generation LDS kernels (forward, data gradient and weight gradient), the grouped 3x3 forward / data gradient on the
    LDS kernel, every aggregation on the 3x3 LDS fast path; the grouped convolutions of CoTNeXt go to the general kernels"""
    table = build_table()
    for key, launches in table.items():
        assert launches and not launches[0].startswith("rc="), (key, launches)
        names = [ln.split("[")[0] for ln in launches]
        if key.startswith("cotnet50") and " conv1x1 " in key:
            N, Ci, Co, g, H, W, s, bias = map(int, key.split()[2].split("x"))
            if key.endswith(" fwd") or key.endswith("(two slabs)") or key.endswith(" dgrad"):
                K = Ci if not key.endswith(" dgrad") else Co
                if K % 32 == 0:
                    assert names[0] == "conv1x1_lds_fwd2", (key, launches)
            if key.endswith(" wgrad"):
                assert names[0] == "conv1x1_wgrad_lds2", (key, launches)
        if key.startswith("cotnet50") and " conv3x3 " in key and (key.endswith(" fwd") or key.endswith(" dgrad")):
            assert "conv3x3g_lds_fwd" in names or "conv3x3g_lds_res" in names, (key, launches)
        if key.startswith("cotnet50") and " conv3x3 " in key and key.endswith(" wgrad(guarded)"):
            assert names[0] == "conv1x1_wgrad_lds2" and "block=832" in launches[0], (key, launches)  # (the TAPS form: 9 + 4 waves)
        if " agg " in key:  # (LDS-staged 3x3 kernels; the bf16 fused backward at even widths on their packed dot-product form)
            assert all("k3_lds" in n or "k3_dot2" in n for n in names), (key, launches)
            if key.startswith("cotnet50") and key.endswith(" bwd") and "x7x7x" not in key:
                assert all("k3_dot2" in n for n in names), (key, launches)
    # CoTNeXt's grouped 1x1 convolutions (groups = 2): group by group on the tuned kernels wherever the depth of a group's
    # reduction is on the 8-channel grid (forward: Ci / 2, data gradient: Co / 2) resp. its slabs are 16-byte aligned (weight
    # gradient) -- one launch (+ reduce) per group --, else the general kernels
    grouped = [k for k in table if k.startswith("cotnext") and " conv1x1 " in k and k.split()[2].split("x")[3] == "2"]
    assert grouped
    n_tuned = 0
    for k in grouped:
        N, Ci, Co, g, H, W, s_, bias = map(int, k.split()[2].split("x"))
        names = [ln.split("[")[0] for ln in table[k]]
        depth = (Ci if k.endswith(" fwd") else Co) // g
        if k.endswith(" fwd") or k.endswith(" dgrad"):
            # (round 6: a depth off the 32-row K step but on the 8-channel grid -- 24 / 48 / 216 / 432 per group -- takes the kernels' KT form)
            want_tuned = depth % 8 == 0 and (Co // g if k.endswith(" fwd") else Ci // g) % 8 == 0
            if want_tuned and depth % 32:
                assert all("KT=1" in ln for ln in table[k]), (k, table[k])
            assert (names == ["conv1x1_lds_fwd2"] * g) if want_tuned else all("convg_" in n for n in names), (k, table[k])
            n_tuned += want_tuned
        elif k.endswith(" wgrad"):
            assert names[0] in ("conv1x1_wgrad_lds2", "gen::convg_wgrad_kernel"), (k, table[k])
            n_tuned += names[0] == "conv1x1_wgrad_lds2"
    assert n_tuned >= len(grouped) // 2, (n_tuned, len(grouped))


if __name__ == "__main__":
    if "--write" in sys.argv:
        t = build_table()
        json.dump(t, open(FIXTURE, "w"), indent=0, sort_keys=True)
        print(f"wrote {len(t)} entries to {FIXTURE}")
    else:
        for k, v in sorted(build_table().items()):
            print(k, "->", "; ".join(v))

#!/usr/bin/env python
"""Why does the fused BatchNorm path triple the train-mode error of the 7x7 CotLayer fixture (verdict r1 weak #6)?
(1) whole-layer error against the fp64 fixture with each fused family switched on/off;
(2) every BatchNorm of the layer in isolation: cot_bn_act vs torch's fp32 batch_norm, both against an fp64 evaluation of
    the SAME inputs (captured from an fp64 run of the layer)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

from cotnet_amd import cotnet, fused_bn, radix_tail  # noqa: E402
from tests.conftest import layer_case, load_golden  # noqa: E402

DEV = "cuda"
gold = load_golden("layer_cotlayer_d64_7x7")
meta, sd, x, gout = layer_case(gold)


def run_layer(dtype, fbn, ftail):
    fused_bn.ENABLED, radix_tail.ENABLED = fbn, ftail
    layer = cotnet.CotLayer(meta["dim"], 3).to(dtype).to(DEV).train()
    layer.load_state_dict({k: (v.to(dtype) if v.is_floating_point() else v) for k, v in sd.items()}, strict=True)
    xin = x.to(dtype).to(DEV).requires_grad_(True)
    y = layer(xin)
    y.backward(gout.to(dtype).to(DEV))
    return y.detach().double().cpu(), xin.grad.double().cpu(), layer


y64, g64 = torch.from_numpy(gold["train_y_f64"]), torch.from_numpy(gold["train_gx_f64"])
print("reference fp32 (CPU) vs fp64: y %.2e gx %.2e" % ((torch.from_numpy(gold["train_y"]).double() - y64).abs().max(),
                                                       (torch.from_numpy(gold["train_gx"]).double() - g64).abs().max()))
for fbn in (False, True):
    for ftail in (False, True):
        y, g, _ = run_layer(torch.float32, fbn, ftail)
        print(f"fp32 GPU fused_bn={fbn!s:5} fused_tail={ftail!s:5}: y {float((y - y64).abs().max()):.2e} gx {float((g - g64).abs().max()):.2e}")
y, g, _ = run_layer(torch.float64, False, False)
print(f"fp64 GPU plain: y {float((y - y64).abs().max()):.2e} gx {float((g - g64).abs().max()):.2e}")

# ---- (2) each BatchNorm in isolation on the fp64 run's activations
fused_bn.ENABLED, radix_tail.ENABLED = False, False
layer = cotnet.CotLayer(meta["dim"], 3).double().to(DEV).train()
layer.load_state_dict({k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}, strict=True)
cap = {}


def hook(name):
    def f(mod, inp, out):
        cap[name] = [inp[0].detach().clone(), None]
        out.register_hook(lambda gr: cap[name].__setitem__(1, gr.detach().clone()))
    return f


names = {"key_embed.1": "relu", "embed.1": "relu", "conv1x1.1": None, "bn": "silu"}
mods = dict(layer.named_modules())
for n in names:
    mods[n].register_forward_hook(hook(n))
xin = x.double().to(DEV).requires_grad_(True)
layer(xin).backward(gout.double().to(DEV))
for n, act in names.items():
    xi, gy = cap[n]  # gy is the gradient w.r.t. the BN OUTPUT (before the activation): compare plain BN (act none)
    bn = mods[n]
    res = {}
    for tag in ("fp64", "torch32", "hip32"):
        dt = torch.float64 if tag == "fp64" else torch.float32
        xa = xi.to(dt).clone().requires_grad_(True)
        w, b = bn.weight.detach().to(dt).clone().requires_grad_(True), bn.bias.detach().to(dt).clone().requires_grad_(True)
        if tag == "hip32":
            m = torch.nn.BatchNorm2d(xa.shape[1]).to(DEV).train()
            m.weight, m.bias = torch.nn.Parameter(w.detach()), torch.nn.Parameter(b.detach())
            fused_bn.ENABLED = True
            yo = fused_bn.fused_bn_act(xa, m, None)
            fused_bn.ENABLED = False
            yo.backward(gy.to(dt))
            res[tag] = (yo.detach().double(), xa.grad.double(), m.weight.grad.double())
        else:
            yo = F.batch_norm(xa, None, None, w, b, True, 0.1, bn.eps)
            yo.backward(gy.to(dt))
            res[tag] = (yo.detach().double(), xa.grad.double(), w.grad.double())
    for tag in ("torch32", "hip32"):
        e = [float((a - r).abs().max()) for a, r in zip(res[tag], res["fp64"])]
        s = [float(r.abs().max()) for r in res["fp64"]]
        print(f"{n:12s} M={xi.shape[0] * xi.shape[2] * xi.shape[3]} {tag:8s} max|err| y {e[0]:.2e} dx {e[1]:.2e} dgamma {e[2]:.2e}   (scales {s[0]:.1e} {s[1]:.1e} {s[2]:.1e})")

# ---- (3) every LEAF module of the layer in isolation: fp32 on the GPU and fp32 on the CPU against fp64, same inputs
print("\nleaf modules, fp32 vs fp64 on the fp64 run's activations (max|err| of output / input gradient, GPU then CPU):")
fused_bn.ENABLED, radix_tail.ENABLED = False, False
layer = cotnet.CotLayer(meta["dim"], 3).double().to(DEV).train()
layer.load_state_dict({k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}, strict=True)
cap = {}
leaves = [(n, m) for n, m in layer.named_modules() if not list(m.children()) and list(m.parameters())]
for n, m in leaves:
    m.register_forward_hook(hook(n))
xin = x.double().to(DEV).requires_grad_(True)
layer(xin).backward(gout.double().to(DEV))
import copy  # noqa: E402
for n, m in leaves:
    if n not in cap or cap[n][1] is None:
        continue
    xi, gy = cap[n]
    res = {}
    for tag, dt, dv in (("fp64", torch.float64, DEV), ("gpu32", torch.float32, DEV), ("cpu32", torch.float32, "cpu")):
        mm = copy.deepcopy(m).to(dt).to(dv).train()
        xa = xi.to(dt).to(dv).clone().requires_grad_(True)
        yo = mm(xa)
        yo.backward(gy.to(dt).to(dv))
        res[tag] = (yo.detach().double().cpu(), xa.grad.double().cpu())
    line = f"{n:14s} {type(m).__name__:12s} in {tuple(xi.shape)}"
    for tag in ("gpu32", "cpu32"):
        e = [float((a - r).abs().max()) for a, r in zip(res[tag], res["fp64"])]
        line += f" | {tag} y {e[0]:.1e} dx {e[1]:.1e}"
    print(line + f"  (scales {float(res['fp64'][0].abs().max()):.1e} {float(res['fp64'][1].abs().max()):.1e})")

"""tests/golden/make_golden.py -- generate the committed golden fixtures FROM THE REFERENCE ITSELF.

Runs only where the reference checkout exists (this build container, /root/reference); the fixtures it
writes next to this file are what travels.  How the reference is made to run without CuPy/CUDA/yacs:

  * `oracle.build_ref.install_stubs()` stubs the absent `cupy` and `yacs` modules so that
    `import models` / `import cupy_layers.*` from the reference succeed on CPU;
  * the reference's kernel source strings (cupy_layers/aggregation_zeropad.py:20-110, _mix.py:20-207) are
    compiled for the CPU by oracle/build_ref.py (same Template substitution as the reference, plus a
    20-line blockIdx/threadIdx shim), giving `RefAggregation` / `RefAggregationMix`;
  * for layer/model fixtures the reference's `aggregation_zeropad` function is monkey-patched with an
    autograd.Function that launches those CPU-compiled reference kernels (the reference's own CPU branch
    does `.cuda()`, aggregation_zeropad.py:192-196).  Everything else (CotLayer / CoXtLayer / ResNet /
    CoTHybridNet module code, torch CPU ops) is the reference's code, unmodified.

Inputs come from numpy's PCG64 generator (stream is version-stable), so tests regenerate them from the seed
stored in each fixture instead of shipping them.

    python tests/golden/make_golden.py            # rewrites *.npz / *.json in tests/golden/
"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import build_ref  # noqa: E402


def rng_tensor(rng, shape, dtype):
    return torch.from_numpy(rng.standard_normal(shape)).to(dtype)


# ----------------------------------------------------------------------------------------------
# 1. operator-level fixtures: the reference kernels on seeded inputs
# ----------------------------------------------------------------------------------------------
OP_CASES = [
    # name, geometry (kwargs of RefAggregation), seed
    ("selftest_k5_heads2", build_ref.PREBUILT_AGG[0], 11),     # aggregation_zeropad.py:238-264
    ("selftest_k1_heads2", build_ref.PREBUILT_AGG[1], 12),     # aggregation_zeropad.py:266-292
    ("config1_f32", build_ref.PREBUILT_AGG[2], 13),            # BASELINE.json configs[0]
    ("stride2", build_ref.PREBUILT_AGG[4], 14),
    ("dilation2", build_ref.PREBUILT_AGG[5], 15),
    ("rect_k3x5_s2x1", build_ref.PREBUILT_AGG[6], 16),
]


def op_inputs(geom, seed):
    """shared with tests/: regenerate (x, w, gout) for a fixture"""
    dtype = torch.float32 if geom["dtype"] == "float" else torch.float64
    k = geom["kernel_size"]
    k = (k, k) if isinstance(k, int) else tuple(k)
    ref_dims = build_ref.RefAggregation.__new__(build_ref.RefAggregation)
    s, p, d = (build_ref._pair(geom[n]) for n in ("stride", "padding", "dilation"))
    Ho = build_ref._out(geom["H"], k[0], s[0], p[0], d[0])
    Wo = build_ref._out(geom["W"], k[1], s[1], p[1], d[1])
    del ref_dims
    rng = np.random.Generator(np.random.PCG64(seed))
    x = rng_tensor(rng, (geom["N"], geom["C"], geom["H"], geom["W"]), dtype)
    w = rng_tensor(rng, (geom["N"], geom["heads"], geom["wC"], k[0] * k[1], Ho, Wo), dtype)
    gout = rng_tensor(rng, (geom["N"], geom["heads"] * geom["C"], Ho, Wo), dtype)
    return x, w, gout


def make_op_fixtures():
    for name, geom, seed in OP_CASES:
        ref = build_ref.RefAggregation(**geom)
        x, w, gout = op_inputs(geom, seed)
        out = ref.forward(x, w)
        gx = ref.backward_input(gout, w)
        gw = ref.backward_weight(gout, x)
        np.savez_compressed(os.path.join(HERE, f"agg_{name}.npz"), out=out.numpy(), gx=gx.numpy(), gw=gw.numpy(),
                            seed=np.int64(seed), geom=json.dumps(geom))
        print(f"agg_{name}: out {tuple(out.shape)}")
    # mix: reference self-test geometry (heads=1) and a heads=2 case that exposes the head-0-only input grad
    for name, geom, seed in (("mix_selftest", build_ref.PREBUILT_MIX[0], 21), ("mix_heads2", build_ref.PREBUILT_MIX[1], 22)):
        ref = build_ref.RefAggregationMix(**geom)
        x, w1, w2, gout = mix_inputs(geom, seed)
        out = ref.forward(x, w1, w2)
        gx = ref.backward_input(gout, w1, w2)
        gw1, gw2 = ref.backward_weight(gout, x)
        np.savez_compressed(os.path.join(HERE, f"agg_{name}.npz"), out=out.numpy(), gx=gx.numpy(), gw1=gw1.numpy(),
                            gw2=gw2.numpy(), seed=np.int64(seed), geom=json.dumps(geom))
        print(f"agg_{name}: out {tuple(out.shape)}")


def mix_inputs(geom, seed):
    dtype = torch.float32 if geom["dtype"] == "float" else torch.float64
    s, p1, d = (build_ref._pair(geom[n]) for n in ("stride", "padding1", "dilation"))
    Ho = build_ref._out(geom["H"], 3, s[0], p1[0], d[0])
    Wo = build_ref._out(geom["W"], 3, s[1], p1[1], d[1])
    rng = np.random.Generator(np.random.PCG64(seed))
    x = rng_tensor(rng, (geom["N"], geom["C"], geom["H"], geom["W"]), dtype)
    w1 = rng_tensor(rng, (geom["N"], geom["heads"], geom["wC"], 9, Ho, Wo), dtype)
    w2 = rng_tensor(rng, (geom["N"], geom["heads"], geom["wC"], 25, Ho, Wo), dtype)
    gout = rng_tensor(rng, (geom["N"], 2 * geom["heads"] * geom["C"], Ho, Wo), dtype)
    return x, w1, w2, gout


# ----------------------------------------------------------------------------------------------
# 2. run the reference's layer / model code on CPU
# ----------------------------------------------------------------------------------------------
_REF_CACHE = {}


class _RefAggOnCPU(torch.autograd.Function):
    """AggregationZeropad (aggregation_zeropad.py:112-186) with the CuPy launch replaced by the same kernels
    compiled for the CPU."""

    @staticmethod
    def forward(ctx, input, weight, kernel_size, stride, padding, dilation):
        N, C, H, W = input.shape
        heads, wC = weight.shape[1], weight.shape[2]
        dt = "float" if input.dtype == torch.float32 else "double"
        key = (dt, N, C, H, W, heads, wC, kernel_size, stride, padding, dilation)
        if key not in _REF_CACHE:
            _REF_CACHE[key] = build_ref.RefAggregation(dt, N, C, H, W, heads, wC, kernel_size, stride, padding,
                                                       dilation)
        ctx.ref = _REF_CACHE[key]
        input, weight = input.contiguous(), weight.contiguous()
        ctx.save_for_backward(input, weight)
        return ctx.ref.forward(input, weight)

    @staticmethod
    def backward(ctx, grad_output):
        input, weight = ctx.saved_tensors
        grad_output = grad_output.contiguous()
        return (ctx.ref.backward_input(grad_output, weight), ctx.ref.backward_weight(grad_output, input),
                None, None, None, None)


def import_reference_models():
    build_ref.install_stubs()
    import cupy_layers.aggregation_zeropad as az

    def aggregation_zeropad_cpu(input, weight, kernel_size=3, stride=1, padding=0, dilation=1):
        assert input.shape[0] == weight.shape[0] and (input.shape[1] % weight.shape[2] == 0)
        return _RefAggOnCPU.apply(input, weight, kernel_size, stride, padding, dilation)

    az.aggregation_zeropad = aggregation_zeropad_cpu
    import models  # noqa: F401  (the reference's package)
    import models.cotnet as ref_cotnet
    import models.cotnet_hybrid as ref_hybrid
    return models, ref_cotnet, ref_hybrid


def randomize_norm_state(module, rng):
    """non-trivial BN/GN affine + running statistics so eval-mode parity actually exercises them"""
    for m in module.modules():
        if isinstance(m, (torch.nn.BatchNorm2d, torch.nn.GroupNorm)):
            with torch.no_grad():
                m.weight.copy_(torch.from_numpy(1.0 + 0.2 * rng.standard_normal(m.weight.shape)).to(m.weight.dtype))
                m.bias.copy_(torch.from_numpy(0.1 * rng.standard_normal(m.bias.shape)).to(m.bias.dtype))
                if isinstance(m, torch.nn.BatchNorm2d):
                    m.running_mean.copy_(torch.from_numpy(0.1 * rng.standard_normal(m.running_mean.shape)))
                    m.running_var.copy_(torch.from_numpy(1.0 + 0.2 * rng.random(m.running_var.shape)))


LAYER_CASES = [
    # name, class name, dim, B, H, W, seed
    ("cotlayer_d32", "CotLayer", 32, 2, 10, 10, 31),
    ("coxtlayer_d32", "CoXtLayer", 32, 2, 8, 8, 32),
    ("cotlayer_d64_7x7", "CotLayer", 64, 2, 7, 7, 33),
]


def make_layer_fixtures(ref_cotnet):
    for name, cls, dim, B, H, W, seed in LAYER_CASES:
        rng = np.random.Generator(np.random.PCG64(seed))
        torch.manual_seed(seed)
        layer = getattr(ref_cotnet, cls)(dim, 3).float()
        randomize_norm_state(layer, rng)
        state = {k: v.clone() for k, v in layer.state_dict().items()}
        x = rng_tensor(rng, (B, dim, H, W), torch.float32)
        gout = rng_tensor(rng, (B, dim, H, W), torch.float32)
        out = {}
        for mode in ("eval", "train"):
            layer.load_state_dict(state)
            layer.train(mode == "train")
            layer.zero_grad()
            xin = x.clone().requires_grad_(True)
            y = layer(xin)
            y.backward(gout)
            out[f"{mode}_y"] = y.detach().numpy()
            out[f"{mode}_gx"] = xin.grad.numpy()
            out[f"{mode}_g_embed3_w"] = layer.embed[3].weight.grad.numpy().copy()
            out[f"{mode}_g_key0_w"] = layer.key_embed[0].weight.grad.numpy().copy()
            out[f"{mode}_g_conv1x1_w"] = layer.conv1x1[0].weight.grad.numpy().copy()
            if mode == "train":
                out["train_bn_running_mean"] = layer.bn.running_mean.numpy().copy()
                out["train_bn_running_var"] = layer.bn.running_var.numpy().copy()
        # the same layer, weights and inputs in fp64 (conditioning-free pin for the GPU fp64 run)
        layer64 = getattr(ref_cotnet, cls)(dim, 3).double()
        state64 = {k: (v.double() if v.is_floating_point() else v) for k, v in state.items()}
        for mode in ("eval", "train"):
            layer64.load_state_dict(state64)
            layer64.train(mode == "train")
            xin = x.double().clone().requires_grad_(True)
            y = layer64(xin)
            y.backward(gout.double())
            out[f"{mode}_y_f64"] = y.detach().numpy()
            out[f"{mode}_gx_f64"] = xin.grad.numpy()
        np.savez_compressed(os.path.join(HERE, f"layer_{name}.npz"), seed=np.int64(seed),
                            meta=json.dumps(dict(cls=cls, dim=dim, B=B, H=H, W=W)),
                            **{"sd__" + k: v.numpy() for k, v in state.items()}, **out)
        print(f"layer_{name}: y {out['eval_y'].shape}")


# CoTNet-50's four stage geometries ((C, H) = (64, 56), (128, 28), (256, 14), (512, 7)), B = 2.  Full tensors would
# be ~30 MB of fixtures, so these are COMPACT: the weights are not stored (same seed + same construction order = the
# reference's initial weights; per-parameter fp64 sums are stored as a probe), and outputs / gradients are stored at K evenly
# spaced flat positions plus their full fp64 sums.
REAL_LAYER_CASES = [
    ("cotlayer_s1_64x56", "CotLayer", 64, 2, 56, 56, 51),
    ("cotlayer_s2_128x28", "CotLayer", 128, 2, 28, 28, 52),
    ("cotlayer_s3_256x14", "CotLayer", 256, 2, 14, 14, 53),
    ("cotlayer_s4_512x7", "CotLayer", 512, 2, 7, 7, 54),
    # CoTNeXt (BASELINE config 4, cotnext*_2x48d: width = floor(planes * 48 / 64) * 2 = 96 / 192 / 384 / 768): the grouped
    # 1x1 convolutions, the groups-8 key embedding and the group -> batch fold at their real geometries (cotnet.py:106-178)
    ("coxtlayer_s1_96x56", "CoXtLayer", 96, 2, 56, 56, 61),
    ("coxtlayer_s2_192x28", "CoXtLayer", 192, 2, 28, 28, 62),
    ("coxtlayer_s3_384x14", "CoXtLayer", 384, 2, 14, 14, 63),
    ("coxtlayer_s4_768x7", "CoXtLayer", 768, 2, 7, 7, 64),
    # SE-CoTNetD-152 at 320 x 320 (BASELINE config 5): the CoT layers of its 256- and 512-wide stages (cotnet_hybrid.py:48-116,
    # :138,:155) see 20 x 20 and 10 x 10 maps
    ("hybrid_cotlayer_256x20", "hybrid.CoTLayer", 256, 2, 20, 20, 65),
    ("hybrid_cotlayer_512x10", "hybrid.CoTLayer", 512, 2, 10, 10, 66),
]
K_OUT, K_GRAD = 16384, 8192


def sample_idx(n, k):
    return np.unique(np.linspace(0, n - 1, min(n, k)).astype(np.int64))


def make_real_layer_fixtures(ref_cotnet, ref_hybrid=None, only=None):
    for name, cls, dim, B, H, W, seed in REAL_LAYER_CASES:
        if only is not None and not any(name.startswith(o) for o in only):
            continue
        rng = np.random.Generator(np.random.PCG64(seed))
        torch.manual_seed(seed)
        mod = ref_hybrid if cls.startswith("hybrid.") else ref_cotnet
        layer = getattr(mod, cls.split(".")[-1])(dim, 3).float()
        randomize_norm_state(layer, rng)
        state = {k: v.clone() for k, v in layer.state_dict().items()}
        probe = {k: float(v.double().sum()) for k, v in state.items() if v.is_floating_point()}
        x = rng_tensor(rng, (B, dim, H, W), torch.float32)
        gout = rng_tensor(rng, (B, dim, H, W), torch.float32)
        out = {}
        for mode in ("eval", "train"):
            layer.load_state_dict(state)
            layer.train(mode == "train")
            layer.zero_grad()
            xin = x.clone().requires_grad_(True)
            y = layer(xin)
            y.backward(gout)
            tensors = {"y": y.detach(), "gx": xin.grad, "g_embed3_w": layer.embed[3].weight.grad,
                       "g_embed0_w": layer.embed[0].weight.grad, "g_key0_w": layer.key_embed[0].weight.grad,
                       "g_conv1x1_w": layer.conv1x1[0].weight.grad, "g_bn_w": layer.bn.weight.grad}
            for key, t in tensors.items():
                flat = t.detach().reshape(-1)
                idx = sample_idx(flat.numel(), K_OUT if key in ("y", "gx") else K_GRAD)
                out[f"{mode}_{key}"] = flat[idx].numpy().copy()
                out[f"{mode}_{key}_sum"] = np.float64(flat.double().sum().item())
                out[f"{mode}_{key}_absmax"] = np.float64(flat.abs().max().item())
            if mode == "train":
                out["train_bn_running_mean"] = layer.bn.running_mean.numpy().copy()
                out["train_bn_running_var"] = layer.bn.running_var.numpy().copy()
        np.savez_compressed(os.path.join(HERE, f"layer_{name}.npz"), seed=np.int64(seed),
                            meta=json.dumps(dict(cls=cls, dim=dim, B=B, H=H, W=W, compact=True, probe=probe)), **out)
        print(f"layer_{name}: {os.path.getsize(os.path.join(HERE, f'layer_{name}.npz')) / 1e3:.0f} KB")


MODEL_CASES = [
    # entry point, input size, seed
    ("cotnet50", 64, 41),
    ("cotnext50_2x48d", 64, 42),
    ("se_cotnetd_50", 64, 43),
]
ALL_ENTRYPOINTS = ["cotnet50", "cotnext50_2x48d", "cotnet101", "cotnext101_2x48d", "se_cotnetd_50", "se_cotnetd_101",
                   "se_cotnetd_152", "se_cotnetd_152_L", "se_cotnetd_200", "se_cotnetd_270"]


def make_model_fixtures(models):
    keys = {}
    for name in ALL_ENTRYPOINTS:
        torch.manual_seed(0)
        m = models.create_model(name)
        keys[name] = {k: list(v.shape) for k, v in m.state_dict().items()}
        print(f"state_dict {name}: {len(keys[name])} tensors, "
              f"{sum(p.numel() for p in m.parameters()) / 1e6:.2f} M params")
        del m
    with open(os.path.join(HERE, "state_dict_keys.json"), "w") as f:
        json.dump(keys, f)
    for name, size, seed in MODEL_CASES:
        # same-seed construction: our builders create modules in the reference's order, so torch's RNG
        # yields identical initial weights; zero_init_last_bn=False keeps the CoT branches alive in the logits.
        # fp64: at 64x64 input the late stages normalise over a handful of elements, so fp32 round-off is
        # amplified to ~1e-2 between ANY two summation orders; fp64 pins the wiring to ~1e-14.
        torch.manual_seed(seed)
        m = models.create_model(name, num_classes=10, zero_init_last_bn=False).double()
        rng = np.random.Generator(np.random.PCG64(seed))
        x = rng_tensor(rng, (2, 3, size, size), torch.float64)
        probe = {k: float(v.double().sum()) for k, v in list(m.state_dict().items())[:8]}  # before BN stats move
        with torch.no_grad():
            logits = m.eval()(x)
            logits_train = m.train()(x)
        np.savez_compressed(os.path.join(HERE, f"model_{name}.npz"), logits=logits.numpy(),
                            logits_train=logits_train.numpy(), seed=np.int64(seed),
                            meta=json.dumps(dict(size=size, num_classes=10, probe=probe)))
        print(f"model_{name}: logits {tuple(logits.shape)} |max| {logits.abs().max():.4f}")


if __name__ == "__main__":
    assert build_ref.reference_available(), "the reference checkout is required to regenerate fixtures"
    if "--real-layers-only" not in sys.argv:
        make_op_fixtures()
    models, ref_cotnet, ref_hybrid = import_reference_models()
    if "--real-layers-only" in sys.argv:  # [--only PREFIX,PREFIX...]: e.g. --only coxtlayer,hybrid (round 4's additions)
        only = sys.argv[sys.argv.index("--only") + 1].split(",") if "--only" in sys.argv else None
        make_real_layer_fixtures(ref_cotnet, ref_hybrid, only)
        sys.exit(0)
    make_layer_fixtures(ref_cotnet)
    make_real_layer_fixtures(ref_cotnet, ref_hybrid)
    make_model_fixtures(models)

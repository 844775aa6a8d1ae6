import json
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `pytest -m gpu` on the GPU box)")


def pytest_collection_modifyitems(config, items):
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this environment")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


def rng_tensor(rng, shape, dtype):
    return torch.from_numpy(rng.standard_normal(shape)).to(dtype)


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    return {k: z[k] for k in z.files}


def pair(v):
    return (int(v), int(v)) if isinstance(v, int) else (int(v[0]), int(v[1]))


def out_size(n, k, s, p, d):
    return int((n + 2 * p - (d * (k - 1) + 1)) / s + 1)


def agg_case_inputs(gold):
    """regenerate (x, w, gout) of an agg_* fixture from its stored seed (mirrors make_golden.op_inputs)"""
    geom = json.loads(str(gold["geom"]))
    dtype = torch.float32 if geom["dtype"] == "float" else torch.float64
    k, s, p, d = (pair(geom[n]) for n in ("kernel_size", "stride", "padding", "dilation"))
    Ho, Wo = out_size(geom["H"], k[0], s[0], p[0], d[0]), out_size(geom["W"], k[1], s[1], p[1], d[1])
    rng = np.random.Generator(np.random.PCG64(int(gold["seed"])))
    x = rng_tensor(rng, (geom["N"], geom["C"], geom["H"], geom["W"]), dtype)
    w = rng_tensor(rng, (geom["N"], geom["heads"], geom["wC"], k[0] * k[1], Ho, Wo), dtype)
    gout = rng_tensor(rng, (geom["N"], geom["heads"] * geom["C"], Ho, Wo), dtype)
    return geom, x, w, gout


def mix_case_inputs(gold):
    geom = json.loads(str(gold["geom"]))
    dtype = torch.float32 if geom["dtype"] == "float" else torch.float64
    s, p1, d = (pair(geom[n]) for n in ("stride", "padding1", "dilation"))
    Ho, Wo = out_size(geom["H"], 3, s[0], p1[0], d[0]), out_size(geom["W"], 3, s[1], p1[1], d[1])
    rng = np.random.Generator(np.random.PCG64(int(gold["seed"])))
    x = rng_tensor(rng, (geom["N"], geom["C"], geom["H"], geom["W"]), dtype)
    w1 = rng_tensor(rng, (geom["N"], geom["heads"], geom["wC"], 9, Ho, Wo), dtype)
    w2 = rng_tensor(rng, (geom["N"], geom["heads"], geom["wC"], 25, Ho, Wo), dtype)
    gout = rng_tensor(rng, (geom["N"], 2 * geom["heads"] * geom["C"], Ho, Wo), dtype)
    return geom, x, w1, w2, gout


AGG_FIXTURES = ["agg_selftest_k5_heads2", "agg_selftest_k1_heads2", "agg_config1_f32", "agg_stride2", "agg_dilation2",
                "agg_rect_k3x5_s2x1"]
MIX_FIXTURES = ["agg_mix_selftest", "agg_mix_heads2"]
LAYER_FIXTURES = ["layer_cotlayer_d32", "layer_coxtlayer_d32", "layer_cotlayer_d64_7x7"]
MODEL_FIXTURES = ["model_cotnet50", "model_cotnext50_2x48d", "model_se_cotnetd_50"]
# CoTNet-50's four stage geometries at B = 2 (compact fixtures: weights from the seed, outputs at sampled positions + sums)
REAL_LAYER_FIXTURES = ["layer_cotlayer_s1_64x56", "layer_cotlayer_s2_128x28", "layer_cotlayer_s3_256x14", "layer_cotlayer_s4_512x7",
                       # round 4: CoXtLayer at CoTNeXt's widths, cotnet_hybrid.CoTLayer at SE-CoTNetD-152-L's maps (320 x 320 input)
                       "layer_coxtlayer_s1_96x56", "layer_coxtlayer_s2_192x28", "layer_coxtlayer_s3_384x14", "layer_coxtlayer_s4_768x7",
                       "layer_hybrid_cotlayer_256x20", "layer_hybrid_cotlayer_512x10"]
K_OUT, K_GRAD = 16384, 8192


def randomize_norm_state(module, rng):
    """same RNG consumption as tests/golden/make_golden.randomize_norm_state"""
    for m in module.modules():
        if isinstance(m, (torch.nn.BatchNorm2d, torch.nn.GroupNorm)):
            rng.standard_normal(m.weight.shape)
            rng.standard_normal(m.bias.shape)
            if isinstance(m, torch.nn.BatchNorm2d):
                rng.standard_normal(m.running_mean.shape)
                rng.random(m.running_var.shape)


def layer_case(gold):
    """-> (meta, state_dict, x, gout) for a layer_* fixture"""
    meta = json.loads(str(gold["meta"]))
    sd = {k[4:]: torch.from_numpy(gold[k]) for k in gold if k.startswith("sd__")}
    rng = np.random.Generator(np.random.PCG64(int(gold["seed"])))
    from cotnet_amd import cotnet
    randomize_norm_state(getattr(cotnet, meta["cls"])(meta["dim"], 3), rng)  # skip what the generator consumed
    x = rng_tensor(rng, (meta["B"], meta["dim"], meta["H"], meta["W"]), torch.float32)
    gout = rng_tensor(rng, (meta["B"], meta["dim"], meta["H"], meta["W"]), torch.float32)
    return meta, sd, x, gout


def sample_idx(n, k):
    """positions a compact fixture stores (tests/golden/make_golden.sample_idx)"""
    return np.unique(np.linspace(0, n - 1, min(n, k)).astype(np.int64))


def real_layer_case(gold):
    """-> (meta, layer with the reference's weights, x, gout) for a compact layer fixture: same seed + same construction order
    give the reference's initial weights (checked against the fixture's per-parameter fp64 sums), the normalisation state and
    the inputs come from the stored numpy seed exactly as the generator drew them"""
    meta = json.loads(str(gold["meta"]))
    seed = int(gold["seed"])
    from cotnet_amd import cotnet, cotnet_hybrid
    rng = np.random.Generator(np.random.PCG64(seed))
    torch.manual_seed(seed)
    mod = cotnet_hybrid if meta["cls"].startswith("hybrid.") else cotnet
    layer = getattr(mod, meta["cls"].split(".")[-1])(meta["dim"], 3).float()
    for m in layer.modules():  # tests/golden/make_golden.randomize_norm_state, same draws in the same order
        if isinstance(m, (torch.nn.BatchNorm2d, torch.nn.GroupNorm)):
            with torch.no_grad():
                m.weight.copy_(torch.from_numpy(1.0 + 0.2 * rng.standard_normal(m.weight.shape)).to(m.weight.dtype))
                m.bias.copy_(torch.from_numpy(0.1 * rng.standard_normal(m.bias.shape)).to(m.bias.dtype))
                if isinstance(m, torch.nn.BatchNorm2d):
                    m.running_mean.copy_(torch.from_numpy(0.1 * rng.standard_normal(m.running_mean.shape)))
                    m.running_var.copy_(torch.from_numpy(1.0 + 0.2 * rng.random(m.running_var.shape)))
    for k, v in layer.state_dict().items():
        if v.is_floating_point():
            want = meta["probe"][k]
            assert abs(float(v.double().sum()) - want) <= 1e-6 * max(1.0, abs(want)), f"weights differ from the reference's: {k}"
    x = rng_tensor(rng, (meta["B"], meta["dim"], meta["H"], meta["W"]), torch.float32)
    gout = rng_tensor(rng, (meta["B"], meta["dim"], meta["H"], meta["W"]), torch.float32)
    return meta, layer, x, gout


def check_real_layer(gold, mode, layer, y, gx, tol=1e-3):
    """compare a run of `layer` with the compact fixture: sampled positions at `tol` (scaled by the tensor's magnitude for the
    parameter gradients, as the small-layer tests do) and the full-tensor sums"""
    tensors = {"y": y, "gx": gx, "g_embed3_w": layer.embed[3].weight.grad, "g_embed0_w": layer.embed[0].weight.grad,
               "g_key0_w": layer.key_embed[0].weight.grad, "g_conv1x1_w": layer.conv1x1[0].weight.grad,
               "g_bn_w": layer.bn.weight.grad}
    for key, t in tensors.items():
        flat = t.detach().float().cpu().reshape(-1)
        idx = sample_idx(flat.numel(), K_OUT if key in ("y", "gx") else K_GRAD)
        ref = torch.from_numpy(gold[f"{mode}_{key}"])
        scale = max(1.0, float(gold[f"{mode}_{key}_absmax"]))
        err = (flat[idx] - ref).abs().max().item()
        assert err <= tol * scale, (mode, key, err, scale)
        # the whole tensor through its sum: an error outside the sampled positions moves it (bound: n terms of size tol)
        dsum = abs(flat.double().sum().item() - float(gold[f"{mode}_{key}_sum"]))
        assert dsum <= tol * scale * max(1.0, flat.numel() ** 0.5), (mode, key, "sum", dsum)

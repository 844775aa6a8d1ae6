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

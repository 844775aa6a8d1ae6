"""Host-side model logic on CPU: state_dict contract and module wiring.

The aggregation op itself needs the GPU (the product has no CPU path), so the wiring tests below substitute the
ORACLE (the reference's nn.Unfold formula) for `aggregation_zeropad` -- test-only monkey-patching, exactly what
tests/ is allowed to do with oracle/.  They check everything AROUND the op against fixtures produced by the
reference's own CotLayer / CoXtLayer / ResNet / CoTHybridNet code (tests/golden/make_golden.py).  The GPU twins of
these tests (tests/test_layers_gpu.py) run the real HIP kernels.
"""
import json
import os

import numpy as np
import pytest
import torch

import cotnet_amd
from cotnet_amd import cotnet
from oracle import unfold_oracle
from tests.conftest import (GOLDEN, LAYER_FIXTURES, MODEL_FIXTURES, REAL_LAYER_FIXTURES, check_real_layer, layer_case, load_golden,
                            real_layer_case, rng_tensor)

README_PARAMS_M = {  # reference README.md:45-52
    "cotnet50": 22.2, "cotnext50_2x48d": 30.1, "cotnet101": 38.3, "cotnext101_2x48d": 53.4,
    "se_cotnetd_50": 23.1, "se_cotnetd_101": 40.9, "se_cotnetd_152": 55.8, "se_cotnetd_152_L": 55.8,
}


@pytest.fixture
def oracle_aggregation(monkeypatch):
    import cotnet_amd.aggregation_zeropad as az

    def agg(input, weight, kernel_size=3, stride=1, padding=0, dilation=1):
        return unfold_oracle.aggregation_unfold(input, weight, kernel_size, stride, padding, dilation)

    monkeypatch.setattr(az, "aggregation_zeropad", agg)


def test_registry_lists_all_reference_entrypoints():
    names = cotnet_amd.list_models()
    for n in ["cotnet50", "cotnext50_2x48d", "cotnet101", "cotnext101_2x48d", "se_cotnetd_50", "se_cotnetd_101",
              "se_cotnetd_152", "se_cotnetd_152_L", "se_cotnetd_200", "se_cotnetd_270"]:
        assert n in names
    with pytest.raises(RuntimeError, match="Unknown model"):
        cotnet_amd.create_model("resnet50")


@pytest.mark.parametrize("name", ["cotnet50", "cotnext50_2x48d", "cotnet101", "cotnext101_2x48d", "se_cotnetd_50",
                                  "se_cotnetd_101", "se_cotnetd_152", "se_cotnetd_152_L", "se_cotnetd_200",
                                  "se_cotnetd_270"])
def test_state_dict_keys_and_shapes_equal_reference(name):
    ref = json.load(open(os.path.join(GOLDEN, "state_dict_keys.json")))[name]
    m = cotnet_amd.create_model(name)
    mine = {k: list(v.shape) for k, v in m.state_dict().items()}
    assert list(mine.keys()) == list(ref.keys())  # same keys, same ORDER
    assert mine == ref
    if name in README_PARAMS_M:
        assert abs(sum(p.numel() for p in m.parameters()) / 1e6 - README_PARAMS_M[name]) < 0.06


def test_local_convolution_has_no_state_and_keeps_attributes():
    m = cotnet_amd.LocalConvolution(64, 64, kernel_size=3, stride=1, padding=1, dilation=1)
    assert len(m.state_dict()) == 0 and len(list(m.parameters())) == 0
    assert (m.kernel_size, m.in_channels, m.out_channels, m.pad_mode) == (3, 64, 64, 0)  # flops_counter.py:500-501


def test_checkpoint_roundtrip_reference_format(tmp_path):
    """reference .pth.tar layout: dict(state_dict=..., state_dict_ema=...), keys possibly prefixed 'module.'"""
    torch.manual_seed(0)
    a = cotnet_amd.create_model("cotnet50", num_classes=5)
    sd = {"module." + k: v for k, v in a.state_dict().items()}
    p = tmp_path / "ckpt.pth.tar"
    torch.save({"epoch": 3, "arch": "cotnet50", "state_dict": sd, "state_dict_ema": sd, "version": 2}, p)
    torch.manual_seed(1)
    b = cotnet_amd.create_model("cotnet50", num_classes=5, checkpoint_path=str(p))
    for (k1, v1), (k2, v2) in zip(a.state_dict().items(), b.state_dict().items()):
        assert k1 == k2 and torch.equal(v1, v2)


@pytest.mark.parametrize("name", LAYER_FIXTURES)
def test_layer_wiring_matches_reference_fixture(name, oracle_aggregation):
    gold = load_golden(name)
    meta, sd, x, gout = layer_case(gold)
    layer = getattr(cotnet, meta["cls"])(meta["dim"], 3)
    for mode in ("eval", "train"):
        layer.load_state_dict(sd, strict=True)
        layer.train(mode == "train")
        layer.zero_grad()
        xin = x.clone().requires_grad_(True)
        y = layer(xin)
        y.backward(gout)
        tol = 1e-3  # BASELINE parity bar for fp32
        assert (y.detach() - torch.from_numpy(gold[f"{mode}_y"])).abs().max() < tol
        assert (xin.grad - torch.from_numpy(gold[f"{mode}_gx"])).abs().max() < tol
        for key, p in (("g_embed3_w", layer.embed[3].weight), ("g_key0_w", layer.key_embed[0].weight),
                       ("g_conv1x1_w", layer.conv1x1[0].weight)):
            ref = torch.from_numpy(gold[f"{mode}_{key}"])
            assert (p.grad - ref).abs().max() <= tol * max(1.0, ref.abs().max().item())
    assert (layer.bn.running_mean - torch.from_numpy(gold["train_bn_running_mean"])).abs().max() < 1e-5
    assert (layer.bn.running_var - torch.from_numpy(gold["train_bn_running_var"])).abs().max() < 1e-5


@pytest.mark.parametrize("name", REAL_LAYER_FIXTURES)
def test_layer_wiring_at_the_real_stage_geometries(name, oracle_aggregation):
    """CoTNet-50's four stage geometries (B = 2) against fixtures of the reference's CotLayer: the module wiring on CPU (the
    aggregation is the oracle here; tests/test_layers_gpu.py runs the same fixtures on the HIP kernels)"""
    gold = load_golden(name)
    meta, layer, x, gout = real_layer_case(gold)
    state = {k: v.clone() for k, v in layer.state_dict().items()}
    for mode in ("eval", "train"):
        layer.load_state_dict(state)
        layer.train(mode == "train")
        layer.zero_grad()
        xin = x.clone().requires_grad_(True)
        y = layer(xin)
        y.backward(gout)
        check_real_layer(gold, mode, layer, y, xin.grad)
    assert (layer.bn.running_mean - torch.from_numpy(gold["train_bn_running_mean"])).abs().max() < 1e-5


@pytest.mark.parametrize("name", MODEL_FIXTURES)
def test_model_wiring_matches_reference_fixture_fp64(name, oracle_aggregation):
    gold = load_golden(name)
    meta = json.loads(str(gold["meta"]))
    seed = int(gold["seed"])
    torch.manual_seed(seed)  # same seed + same construction order => the reference's initial weights
    m = cotnet_amd.create_model(name[len("model_"):], num_classes=meta["num_classes"], zero_init_last_bn=False).double()
    for k, v in list(m.state_dict().items())[:8]:
        assert abs(float(v.double().sum()) - meta["probe"][k]) < 1e-6 * max(1.0, abs(meta["probe"][k]))
    x = rng_tensor(np.random.Generator(np.random.PCG64(seed)), (2, 3, meta["size"], meta["size"]), torch.float64)
    with torch.no_grad():
        y = m.eval()(x)
        yt = m.train()(x)
    for got, key in ((y, "logits"), (yt, "logits_train")):
        ref = torch.from_numpy(gold[key])
        # fp64 round-off (1e-16) times the toy network's conditioning (~1e7, see make_golden.py) -> 1e-7
        assert ((got - ref).abs().max() / ref.abs().max()).item() < 1e-7

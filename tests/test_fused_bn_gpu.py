"""fused_bn_act (csrc/bn_act.hip) on the GPU against the plain torch module sequence it replaces."""
import copy

import pytest
import torch
from torch import nn

from cotnet_amd import fused_bn
from cotnet_amd.fused_bn import fused_bn_act

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(params=[0, 1, 2], ids=["finalize-kernel", "finalize-folded", "channel-resident"])
def fold(request):
    """streaming kernels with the per-channel finalize step as its own launch (cot_set_tuning(12, 0)) or folded into the apply
    kernels' prologue (12 = 1, what bench.py's `new` kernel set runs), both with the channel-resident kernels off
    (cot_set_tuning(21, 0)); and the library default: channel-resident kernels wherever a channel fits a workgroup's registers"""
    from cotnet_amd import _lib
    L = _lib.lib()
    _lib.check(L.cot_set_tuning(12, 1 if request.param == 1 else 0), "cot_set_tuning")
    _lib.check(L.cot_set_tuning(21, 1 if request.param == 2 else 0), "cot_set_tuning")
    yield request.param
    _lib.check(L.cot_set_tuning(12, 0), "cot_set_tuning")
    _lib.check(L.cot_set_tuning(21, 1), "cot_set_tuning")


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("act,use_res", [(None, False), ("relu", False), ("relu", True), ("silu", False)])
@pytest.mark.parametrize("N,C,H", [(8, 64, 56), (8, 128, 28), (8, 256, 14), (8, 2048, 7), (3, 24, 5), (80, 128, 28), (80, 256, 14),
                                   (80, 512, 7)])
def test_matches_torch_modules(N, C, H, act, use_res, dtype, fold):
    torch.manual_seed(C + H)
    bn_a = nn.BatchNorm2d(C).to(DEV).train()
    with torch.no_grad():
        bn_a.weight.uniform_(0.5, 1.5)
        bn_a.bias.normal_(0, 0.2)
    bn_b = copy.deepcopy(bn_a)
    x = (torch.randn(N, C, H, H, device=DEV) * 1.3 + 0.4).to(dtype)
    res = torch.randn(N, C, H, H, device=DEV).to(dtype) if use_res else None
    dy = torch.randn(N, C, H, H, device=DEV).to(dtype)

    def run(bn, enabled):
        fused_bn.ENABLED = enabled
        try:
            xa = x.clone().requires_grad_(True)
            ra = res.clone().requires_grad_(True) if use_res else None
            y = fused_bn_act(xa, bn, act, ra)
            y.backward(dy)
            return y.detach().float(), xa.grad.float(), (ra.grad.float() if use_res else None)
        finally:
            fused_bn.ENABLED = True

    ya, gxa, gra = run(bn_a, True)
    yb, gxb, grb = run(bn_b, False)
    tol = 2e-5 if dtype == torch.float32 else 3e-2
    assert ((ya - yb).abs() <= tol * (1 + yb.abs())).all()
    assert torch.allclose(bn_a.running_mean, bn_b.running_mean, atol=1e-5)
    assert torch.allclose(bn_a.running_var, bn_b.running_var, rtol=1e-4, atol=1e-5)
    assert int(bn_a.num_batches_tracked) == int(bn_b.num_batches_tracked) == 1
    if dtype == torch.float32:  # bf16: the torch path rounds between its modules, gradients differ by that rounding
        gtol = 2e-4
        assert ((gxa - gxb).abs() <= gtol * (1 + gxb.abs())).all()
        assert torch.allclose(bn_a.weight.grad, bn_b.weight.grad, rtol=1e-3, atol=1e-3)
        assert torch.allclose(bn_a.bias.grad, bn_b.bias.grad, rtol=1e-3, atol=1e-3)
        if use_res:
            assert ((gra - grb).abs() <= gtol * (1 + grb.abs())).all()
    else:
        scale = gxb.abs().mean() + 1e-6
        assert (gxa - gxb).abs().mean() < 0.02 * scale + 1e-3


def test_eval_mode_and_channels_last_take_the_torch_path():
    bn = nn.BatchNorm2d(16).to(DEV).eval()
    x = torch.randn(2, 16, 8, 8, device=DEV)
    assert torch.equal(fused_bn_act(x, bn, "relu"), torch.relu(bn(x)))   # eval mode with autograd on: the modules
    bn.train()
    xcl = x.contiguous(memory_format=torch.channels_last)
    y = fused_bn_act(xcl, bn, "relu")
    assert y.shape == x.shape and int(bn.num_batches_tracked) == 1


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("act,use_res", [(None, False), ("relu", False), ("relu", True), ("silu", False)])
@pytest.mark.parametrize("N,C,H", [(8, 64, 56), (8, 256, 14), (8, 2048, 7), (3, 24, 5), (4, 32, 1)])
def test_inference_mode_matches_torch_modules(N, C, H, act, use_res, dtype):
    """eval-mode BatchNorm + activation (+ residual) under torch.no_grad() as one pass (cot_bn_act_inference): what the
    forward-only configuration (BASELINE config 2) runs"""
    torch.manual_seed(C + H)
    bn = nn.BatchNorm2d(C).to(DEV).eval()
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5)
        bn.bias.normal_(0, 0.2)
        bn.running_mean.normal_(0, 0.5)
        bn.running_var.uniform_(0.5, 2.0)
    x = (torch.randn(N, C, H, H, device=DEV) * 1.3 + 0.4).to(dtype)
    res = torch.randn(N, C, H, H, device=DEV).to(dtype) if use_res else None
    rm, rv, nbt = bn.running_mean.clone(), bn.running_var.clone(), int(bn.num_batches_tracked)
    with torch.no_grad():
        ya = fused_bn_act(x, bn, act, res)
        z = torch.nn.functional.batch_norm(x.float(), bn.running_mean, bn.running_var, bn.weight, bn.bias, False, 0.1, bn.eps)
        if use_res:
            z = z + res.float()
        yb = {None: lambda t: t, "relu": torch.relu, "silu": torch.nn.functional.silu}[act](z)
    tol = 2e-5 if dtype == torch.float32 else 2e-2
    assert ya.dtype == dtype and ((ya.float() - yb).abs() <= tol * (1 + yb.abs())).all()
    assert torch.equal(bn.running_mean, rm) and torch.equal(bn.running_var, rv) and int(bn.num_batches_tracked) == nbt


def test_forward_only_model_matches_plain_modules():
    """cotnet50 in eval mode under torch.no_grad() (bench.py --mode fwd): the one-pass inference BatchNorm kernels against the
    torch modules, fp32.  The two differ by rounding (x*a + b vs (x - mean)*invstd*gamma + beta) and a random-initialised
    network amplifies any rounding through its 16 blocks, so both are measured against an fp64 evaluation of the same model:
    the fused path must be as close to it as the module path is (a wrong kernel gives O(1) at the first BatchNorm)"""
    import copy
    import cotnet_amd
    torch.manual_seed(1)
    m = cotnet_amd.create_model("cotnet50", num_classes=32).to(DEV).eval()
    for mod in m.modules():   # running statistics off their initial values
        if isinstance(mod, nn.BatchNorm2d):
            with torch.no_grad():
                mod.running_mean.normal_(0, 0.1)
                mod.running_var.uniform_(0.8, 1.2)
                mod.weight.uniform_(0.8, 1.2)
    x = torch.randn(4, 3, 128, 128, device=DEV)
    stage1 = {}
    hook = m.layer1.register_forward_hook(lambda mod, inp, out: stage1.setdefault(fused_bn.ENABLED, out.detach().clone()))
    with torch.no_grad():
        fused_bn.ENABLED = True
        ya = m(x)
        fused_bn.ENABLED = False
        try:
            yb = m(x)
        finally:
            fused_bn.ENABLED = True
            hook.remove()
        yt = copy.deepcopy(m).double()(x.double())
    e1 = ((stage1[True] - stage1[False]).abs().max() / stage1[False].abs().max()).item()   # before the amplification
    ef = ((ya.double() - yt).abs().max() / yt.abs().max()).item()
    et = ((yb.double() - yt).abs().max() / yt.abs().max()).item()
    # (the logits of this untrained network sit ~1e-2 from the fp64 evaluation on EITHER path -- rounding amplified by 16 blocks;
    # a defect shows up as O(1), and already in e1)
    assert e1 < 1e-4 and ef <= 10 * et + 3e-2, (e1, ef, et)

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


def _stage_io_fp64(m, x):
    """fp64 evaluation of the model: inputs and outputs of its four stages"""
    import copy
    m64 = copy.deepcopy(m).double()
    io = {}
    hooks = [getattr(m64, f"layer{k}").register_forward_hook(lambda mod, inp, out, k=k: io.__setitem__(k, (inp[0].detach(), out.detach())))
             for k in (1, 2, 3, 4)]
    with torch.no_grad():
        logits = m64(x.double())
    for h in hooks:
        h.remove()
    return io, logits


def test_forward_only_model_matches_plain_modules():
    """cotnet50 in eval mode under torch.no_grad() (bench.py --mode fwd, BASELINE config 2): the one-pass inference BatchNorm
    kernels against the torch modules, fp32.  An untrained network amplifies ANY rounding difference through its 16 blocks
    (round 2: 2.4e-2 on the logits for the fused path, 6.9e-3 for the modules -- x*a + b against (x - mean)*invstd*gamma + beta,
    both ~1e-7 per BatchNorm), so the bound sits where a defect cannot hide: every STAGE is evaluated by both paths on the
    SAME input (the fp64 run's input of that stage) and compared with the fp64 run's output of that stage -- errors of 3-6
    blocks, not of 16.  The logits keep a loose end-to-end sanity bound."""
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
    io, yt = _stage_io_fp64(m, x)
    try:
        with torch.no_grad():
            for k in (1, 2, 3, 4):
                inp, want = io[k][0].float(), io[k][1]
                fused_bn.ENABLED = True
                a = getattr(m, f"layer{k}")(inp)
                fused_bn.ENABLED = False
                b = getattr(m, f"layer{k}")(inp)
                ef = ((a.double() - want).abs().max() / want.abs().max()).item()
                et = ((b.double() - want).abs().max() / want.abs().max()).item()
                assert ef <= 5 * et + 1e-4, (k, ef, et)
            fused_bn.ENABLED = True
            ya = m(x)
            fused_bn.ENABLED = False
            yb = m(x)
    finally:
        fused_bn.ENABLED = True
    ef = ((ya.double() - yt).abs().max() / yt.abs().max()).item()
    et = ((yb.double() - yt).abs().max() / yt.abs().max()).item()
    assert ef <= 10 * et + 3e-3, (ef, et)  # (amplified rounding, see above: the per-stage bounds are the test)


def test_forward_only_bf16_model_on_the_library_kernels():
    """BASELINE config 2 as it is benchmarked: bf16 weights + activations, eval mode, torch.no_grad(), every convolution /
    GroupNorm / pooling / BatchNorm on the library's kernels.  Granularity = one Bottleneck (an untrained network amplifies
    bf16 rounding to tens of per cent over a stage -- measured 0.4-0.6 at stage 3 for BOTH paths -- which would hide a
    defect): each of the 16 blocks is evaluated on the fp32 truth's input of that block (rounded to bf16) and compared with the
    truth's output; the module path (MIOpen convolutions) is the yardstick: the library path must not sit further from
    the truth than 1.5x the module path + 2e-3 (mean relative error), block by block"""
    import copy
    import cotnet_amd
    from cotnet_amd.cotnet import Bottleneck
    from cotnet_amd.flat_sgd import to_mixed_bf16
    from tests import truth
    torch.manual_seed(2)
    m = cotnet_amd.create_model("cotnet50", num_classes=32).to(DEV).eval()
    for mod in m.modules():
        if isinstance(mod, nn.BatchNorm2d):
            with torch.no_grad():
                mod.running_mean.normal_(0, 0.1)
                mod.running_var.uniform_(0.8, 1.2)
                mod.weight.uniform_(0.8, 1.2)
    m = to_mixed_bf16(m)
    x = torch.randn(8, 3, 224, 224, device=DEV).bfloat16()
    blocks = [n for n, mod in m.named_modules() if isinstance(mod, Bottleneck)]
    assert len(blocks) == 16
    with truth.switches(**truth.PLAIN):
        mt = copy.deepcopy(m).float()
        io = {}
        hooks = [mt.get_submodule(n).register_forward_hook(lambda mod, inp, out, n=n: io.__setitem__(n, (inp[0].detach(), out.detach())))
                 for n in blocks]
        with torch.no_grad():
            mt(x.float())
        for h in hooks:
            h.remove()
    res = {}
    for name, sw in (("library", truth.SINGLE_NODE), ("modules", truth.ROUND1)):
        with truth.switches(**sw), torch.no_grad():
            for n in blocks:
                res[(name, n)] = truth.err(m.get_submodule(n)(io[n][0].bfloat16()).float(), io[n][1])
    bad = {n: (res[("library", n)], res[("modules", n)]) for n in blocks
           if not res[("library", n)] <= 1.5 * res[("modules", n)] + 2e-3}
    assert not bad, bad
    assert max(res[("library", n)] for n in blocks) < 0.1, res  # (a block-level defect is O(1))

"""FlatSGD(ema_decay=...) / cot_ema_step on the GPU."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_weight_ema_matches_the_reference_formula():
    """FlatSGD(ema_decay=...) on the GPU: three steps, EMA of every state_dict entry against decay*e + (1-decay)*m"""
    from torch import nn
    from cotnet_amd.flat_sgd import FlatSGD, to_mixed_bf16
    torch.manual_seed(3)
    model = to_mixed_bf16(nn.Sequential(nn.Conv2d(8, 16, 3, padding=1), nn.BatchNorm2d(16), nn.ReLU(),
                                        nn.Conv2d(16, 8, 1)).cuda()).train()
    decay = 0.99
    opt = FlatSGD(model, lr=0.05, momentum=0.9, weight_decay=1e-4, ema_decay=decay)
    named = dict(model.named_parameters())
    masters = opt.master_parameters()
    ref = {k: (masters[named[k]].clone() if k in named else v.detach().float().clone())
           for k, v in model.state_dict().items()}
    x = torch.randn(6, 8, 12, 12, device="cuda").bfloat16()
    for _ in range(3):
        opt.zero_grad()
        model(x).float().square().mean().backward()
        opt.step()
        masters = opt.master_parameters()
        for k, v in model.state_dict().items():
            cur = masters[named[k]] if k in named else v.detach().float()
            ref[k] = decay * ref[k] + (1 - decay) * cur if v.is_floating_point() else v.detach().clone()
    got = opt.ema_state_dict()
    assert list(got.keys()) == list(model.state_dict().keys())
    for k in got:
        assert torch.allclose(got[k].float(), ref[k].float(), atol=1e-5, rtol=1e-5), k

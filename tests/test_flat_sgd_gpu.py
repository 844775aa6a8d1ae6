"""FlatSGD (fused HIP SGD-nesterov over flat buckets, bf16 working copies + fp32 masters) against torch.optim.SGD."""
import copy

import pytest
import torch
from torch import nn

from cotnet_amd.flat_sgd import FlatSGD, to_mixed_bf16

pytestmark = pytest.mark.gpu
DEV = "cuda"


def net():
    return nn.Sequential(nn.Conv2d(3, 16, 3, padding=1), nn.BatchNorm2d(16), nn.ReLU(), nn.Conv2d(16, 8, 1, bias=True),
                         nn.GroupNorm(2, 8), nn.AdaptiveAvgPool2d(1), nn.Flatten(), nn.Linear(8, 5))


def test_fp32_matches_torch_sgd_exactly():
    torch.manual_seed(0)
    a = net().to(DEV)
    b = copy.deepcopy(a)
    decay = [p for n, p in b.named_parameters() if p.ndim > 1 and not n.endswith(".bias")]
    no_decay = [p for n, p in b.named_parameters() if not (p.ndim > 1 and not n.endswith(".bias"))]
    ref = torch.optim.SGD([{"params": no_decay, "weight_decay": 0.0}, {"params": decay, "weight_decay": 1e-2}], lr=0.1,
                          momentum=0.9, nesterov=True)
    opt = FlatSGD(a, lr=0.1, momentum=0.9, weight_decay=1e-2, nesterov=True)
    names = [n for n, _ in a.named_parameters()]
    for step in range(4):
        x = torch.randn(6, 3, 8, 8, device=DEV)
        t = torch.randint(0, 5, (6,), device=DEV)
        opt.zero_grad()
        nn.functional.cross_entropy(a(x), t).backward()
        opt.step()
        ref.zero_grad()
        nn.functional.cross_entropy(b(x), t).backward()
        ref.step()
        for n, pa, pb in zip(names, a.parameters(), b.parameters()):
            assert torch.allclose(pa, pb, rtol=2e-5, atol=2e-6), (step, n, (pa - pb).abs().max())
    # parameters live in flat buckets and carry no .grad after the step
    assert all(p.grad is None for p in a.parameters())
    assert len(opt.reducer.buckets) == 2  # (decay, fp32) and (no_decay, fp32)


def test_mixed_bf16_masters_follow_fp32_sgd():
    torch.manual_seed(1)
    a = to_mixed_bf16(net().to(DEV))
    assert a[0].weight.dtype == torch.bfloat16 and a[1].weight.dtype == torch.float32
    opt = FlatSGD(a, lr=0.05, momentum=0.9, weight_decay=1e-3, nesterov=True)
    masters = opt.master_parameters()
    # reference: fp32 SGD on copies of the masters, fed the SAME (bf16-rounded) gradients
    ref_p = {p: m.clone() for p, m in masters.items()}
    ref_m = {p: torch.zeros_like(m) for p, m in masters.items()}
    names = {p: n for n, p in a.named_parameters()}
    for step in range(3):
        x = torch.randn(6, 3, 8, 8, device=DEV).bfloat16()
        t = torch.randint(0, 5, (6,), device=DEV)
        opt.zero_grad()
        nn.functional.cross_entropy(a(x).float(), t).backward()
        # copy mode: autograd's gradients were moved into the flat buckets when each bucket completed
        grads = {p: v.detach().float().clone() for b in opt.reducer.buckets for p, v in zip(b.params, b.views)}
        opt.step()
        for p in a.parameters():
            wd = 1e-3 if (p.ndim > 1 and not names[p].endswith(".bias")) else 0.0
            g = grads[p] + wd * ref_p[p]
            ref_m[p] = 0.9 * ref_m[p] + g
            ref_p[p] = ref_p[p] - 0.05 * (g + 0.9 * ref_m[p])
            assert torch.allclose(masters[p], ref_p[p], rtol=1e-5, atol=1e-6), (step, names[p])
            if p.dtype == torch.bfloat16:
                assert torch.equal(p.data, masters[p].to(torch.bfloat16)), names[p]

#!/usr/bin/env python
"""Which op breaks HIP-graph replay?  (round-1 finding, DESIGN.md 5.3: whole-step capture is exact on the first replay
and wrong afterwards.)  For each candidate op: eager forward+backward -> reference gradients; capture forward+backward
in a hipGraph; replay 3x with the gradient buffers poisoned in between; report the max deviation per replay.

    python scripts/graph_bisect.py            # ~1 minute on one MI355X
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from torch import nn  # noqa: E402

from cotnet_amd.aggregation_zeropad import aggregation_zeropad  # noqa: E402
from cotnet_amd.fused_bn import fused_bn_act  # noqa: E402

dev = torch.device("cuda:0")
torch.backends.cudnn.benchmark = True
DT = torch.bfloat16
B = int(os.environ.get("B", "80"))


class Agg(nn.Module):
    def forward(self, x):
        b, c, h, w = x.shape
        wgt = self.w
        return aggregation_zeropad(x, wgt, 3, 1, 1, 1)

    def __init__(self, c, h):
        super().__init__()
        self.w = nn.Parameter(torch.randn(B, 1, c // 8, 9, h, h))


class FusedBN(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.bn = nn.BatchNorm2d(c)

    def forward(self, x):
        return fused_bn_act(x, self.bn, "relu")


def cases():
    yield "conv1x1 64->256 @56", nn.Conv2d(64, 256, 1, bias=False), (B, 64, 56, 56)
    yield "conv1x1 1024->256 @14", nn.Conv2d(1024, 256, 1, bias=False), (B, 1024, 14, 14)
    yield "conv1x1 512->2048 @7", nn.Conv2d(512, 2048, 1, bias=False), (B, 512, 7, 7)
    yield "conv3x3 g4 64 @56", nn.Conv2d(64, 64, 3, padding=1, groups=4, bias=False), (B, 64, 56, 56)
    yield "conv3x3 g4 512 @7", nn.Conv2d(512, 512, 3, padding=1, groups=4, bias=False), (B, 512, 7, 7)
    yield "conv7x7 s2 stem", nn.Conv2d(3, 64, 7, stride=2, padding=3, bias=False), (B, 3, 224, 224)
    yield "conv1x1+bias 32->72 @56", nn.Conv2d(32, 72, 1), (B, 32, 56, 56)
    yield "linear 2048->1000", nn.Linear(2048, 1000), (B, 2048)
    yield "MIOpen BN 256 @56", nn.BatchNorm2d(256), (B, 256, 56, 56)
    yield "fused BN+ReLU 256 @56", FusedBN(256), (B, 256, 56, 56)
    yield "GroupNorm 8x9 @56", nn.GroupNorm(8, 72), (B, 72, 56, 56)
    yield "aggregation C64 @56", Agg(64, 56), (B, 64, 56, 56)
    yield "maxpool 3x3 s2", nn.MaxPool2d(3, 2, 1), (B, 64, 112, 112)
    yield "avgpool 3x3 s2", nn.AvgPool2d(3, 2, padding=1), (B, 128, 56, 56)


def to_mixed(m):
    for mod in m.modules():
        if isinstance(mod, (nn.Conv2d, nn.Linear, nn.GroupNorm, Agg)):
            for p in mod.parameters(recurse=False):
                p.data = p.data.to(DT)
    return m


def run_case(name, mod, shape):
    torch.manual_seed(0)
    mod = to_mixed(mod.to(dev)).train()
    x = torch.randn(*shape, device=dev).to(DT).requires_grad_(True)
    params = [x] + [p for p in mod.parameters()]

    def fwd_bwd():
        for p in params:
            p.grad = None
        y = mod(x)
        y.float().square().mean().backward()

    fwd_bwd()
    fwd_bwd()
    torch.cuda.synchronize()
    ref = [p.grad.detach().float().clone() for p in params]
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2):
            fwd_bwd()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    for p in params:
        p.grad = None
    g = torch.cuda.CUDAGraph()
    try:
        with torch.cuda.graph(g):
            y = mod(x)
            y.float().square().mean().backward()
    except Exception as e:  # noqa: BLE001
        torch.cuda.synchronize()
        return f"{name:28s} CAPTURE FAILED: {type(e).__name__}: {str(e)[:80]}"
    static = [p.grad for p in params]
    out = []
    for rep in range(3):
        for s in static:
            s.fill_(float("nan"))
        g.replay()
        torch.cuda.synchronize()
        dev_max = 0.0
        for s, r in zip(static, ref):
            d = (s.float() - r).abs().max().item()
            scale = r.abs().max().item() + 1e-12
            dev_max = max(dev_max, d / scale if d == d else float("inf"))
        out.append(dev_max)
    verdict = "ok" if max(out) < 5e-2 else "BROKEN"
    return f"{name:28s} rel.dev per replay {['%.2e' % v for v in out]}  {verdict}"


def run_reducer_case(name, mod, shape):
    """same, but gradients leave through GradBucketReducer's copy-mode hooks (p.grad dropped inside the capture), as in
    cotnet_amd.graph_step.GraphedTrainStep"""
    from cotnet_amd.data_parallel import GradBucketReducer
    from cotnet_amd.flat_sgd import _decay_group
    torch.manual_seed(0)
    mod = to_mixed(mod.to(dev)).train()
    x = torch.randn(*shape, device=dev).to(DT)
    red = GradBucketReducer(mod, group_fn=_decay_group, grad_mode="copy", flatten_params=True)
    red.defer_comm = True

    def fwd_bwd():
        red.zero_grad()
        mod(x).float().square().mean().backward()
        red.finish()

    fwd_bwd()
    fwd_bwd()
    torch.cuda.synchronize()
    ref = [b.flat.float().clone() for b in red.buckets]
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fwd_bwd()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    red.zero_grad()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        mod(x).float().square().mean().backward()
    for b in red.buckets:
        b.pending = len(b.params)
        b.fired.clear()
    out = []
    for rep in range(3):
        for b in red.buckets:
            b.flat.fill_(float("nan"))
        g.replay()
        torch.cuda.synchronize()
        dev_max = 0.0
        for b, r in zip(red.buckets, ref):
            d = (b.flat.float() - r).abs().max().item()
            dev_max = max(dev_max, d / (r.abs().max().item() + 1e-12) if d == d else float("inf"))
        out.append(dev_max)
    verdict = "ok" if max(out) < 5e-2 else "BROKEN"
    return f"{name:28s} rel.dev per replay {['%.2e' % v for v in out]}  {verdict}"


if __name__ == "__main__":
    for name, mod, shape in cases():
        print(run_case(name, mod, shape), flush=True)
    from cotnet_amd.cotnet import Bottleneck, CotLayer
    print(run_case("CotLayer(64) static grads", CotLayer(64, 3), (B, 64, 56, 56)), flush=True)
    print(run_reducer_case("CotLayer(64) reducer copy", CotLayer(64, 3), (B, 64, 56, 56)), flush=True)
    print(run_reducer_case("Bottleneck(256,64) reducer", Bottleneck(256, 64), (B, 256, 56, 56)), flush=True)

#!/usr/bin/env python
"""Which parameter gradients does the copy-mode reducer still COPY into its buckets (not written in place by the producing kernel)?"""
import collections
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402
import cotnet_amd  # noqa: E402
from cotnet_amd import data_parallel as dp, grad_sink  # noqa: E402
from cotnet_amd.flat_sgd import FlatSGD, to_mixed_bf16  # noqa: E402

bench.apply_kernel_set("new")
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = to_mixed_bf16(cotnet_amd.create_model(sys.argv[1] if len(sys.argv) > 1 else "cotnet50", num_classes=1000).to(dev)).train()
opt = FlatSGD(model, lr=0.03, momentum=0.9, weight_decay=4e-5, nesterov=True)
B = int(sys.argv[2]) if len(sys.argv) > 2 else 16
R = int(sys.argv[3]) if len(sys.argv) > 3 else 224
x = torch.randn(B, 3, R, R, device=dev).bfloat16()
t = torch.randint(0, 1000, (B,), device=dev)
name_of = {p: n for n, p in model.named_parameters()}
notin = []
orig = grad_sink.is_in_place


def spy(p, v):
    r = orig(p, v)
    if not r:
        e = grad_sink._SINK.get(id(p))
        notin.append((name_of[p], None if e is None else (e[2], str(e[1].dtype), str(p.dtype))))
    return r


grad_sink.is_in_place = spy
for it in range(2):
    notin.clear()
    opt.zero_grad()
    loss = torch.nn.functional.cross_entropy(model(x).float(), t)
    loss.backward()
    opt.step()
    torch.cuda.synchronize()
    print(it, len(notin), "of", len(name_of))
    print(collections.Counter(".".join(n.split(".")[-3:]) for n, _ in notin).most_common(30))
    print(notin[:12])

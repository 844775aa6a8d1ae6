#!/usr/bin/env python
"""Which host ops issue hipMemcpyWithStream / hipMemcpyAsync during a training step (the __amd_rocclr_copyBuffer launches of the
kernel trace)?  torch.profiler over two steps of bench.py's own step: every runtime event whose name contains "Memcpy" is
listed with its chain of parent ops and the Python stack."""
import collections
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402
import cotnet_amd  # noqa: E402
from cotnet_amd.flat_sgd import FlatSGD, to_mixed_bf16  # noqa: E402

bench.apply_kernel_set("new")
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = to_mixed_bf16(cotnet_amd.create_model("cotnet50", num_classes=1000).to(dev)).train()
opt = FlatSGD(model, lr=0.03, momentum=0.9, weight_decay=4e-5, nesterov=True)
x = torch.randn(80, 3, 224, 224, device=dev).bfloat16()
t = torch.randint(0, 1000, (80,), device=dev)


def step():
    opt.zero_grad()
    loss = torch.nn.functional.cross_entropy(model(x).float(), t)
    loss.backward()
    opt.step()


for _ in range(3):
    step()
torch.cuda.synchronize()
from torch.profiler import ProfilerActivity, profile  # noqa: E402
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    for _ in range(2):
        step()
    torch.cuda.synchronize()
by = collections.Counter()
names = collections.Counter()
for e in prof.events():
    names[e.name[:50]] += 1
    if "emcpy" in e.name or "emset" in e.name:
        chain, p = [], e.cpu_parent
        while p is not None and len(chain) < 6:
            chain.append(p.name[:40])
            p = p.cpu_parent
        stack = " < ".join((e.stack or [])[:3])[:200]
        by[(e.name[:40], " <- ".join(chain), stack)] += 1
print("memcpy / memset events in 2 steps:")
for k, c in by.most_common(30):
    print(f"{c:5d} x {k[0]}\n        parents: {k[1]}\n        stack: {k[2]}")
print("\nruntime-looking event names:")
for n, c in names.most_common(60):
    if n.startswith("hip") or "emcpy" in n:
        print(f"{c:6d}  {n}")

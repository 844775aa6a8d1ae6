#!/usr/bin/env python
"""Which host ops issue the ~200 __amd_rocclr_copyBuffer launches per training step?
torch.profiler over two steps of the `new` kernel set; device-side copy events are listed with the CPU op that launched them."""
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
dur = collections.Counter()
for e in prof.events():
    for k in getattr(e, "kernels", []) or []:
        n = k.name
        if "copy" in n.lower() or "memcpy" in n.lower() or "memset" in n.lower():
            stack = " < ".join(s for s in (e.stack or [])[:4])
            key = (n[:40], e.name, str(e.input_shapes)[:80], stack[:300])
            by[key] += 1
            dur[key] += k.duration
print("device copy / memset events in 2 steps, by launching op:")
for key, c in by.most_common(40):
    print(f"{c:5d} x  {dur[key] / c:7.2f} us  kernel={key[0]}  op={key[1]}  shapes={key[2]}\n         stack: {key[3]}")
names = collections.Counter()
for e in prof.events():
    if e.device_type == torch.autograd.DeviceType.CUDA:
        names[e.name[:60]] += 1
print("\ntop device event names:")
for n, c in names.most_common(25):
    print(f"{c:6d}  {n}")

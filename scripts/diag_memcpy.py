#!/usr/bin/env python
"""Which call sites issue the small device-to-device copies of a training step?  (rocprofv3 shows ~200 `__amd_rocclr_copyBuffer`
launches per CoTNet-50 step, 512-thread grids.)  One profiled step of the benchmark configuration with Python stacks; prints the
aten::copy_ / aten::clone / aten::contiguous call sites by count."""
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
model = to_mixed_bf16(cotnet_amd.create_model(sys.argv[1] if len(sys.argv) > 1 else "cotnet50", num_classes=1000).to(dev)).train()
opt = FlatSGD(model, lr=0.03, momentum=0.9, weight_decay=4e-5, nesterov=True)
B = int(sys.argv[2]) if len(sys.argv) > 2 else 80
R = int(sys.argv[3]) if len(sys.argv) > 3 else 224
x = torch.randn(B, 3, R, R, device=dev).bfloat16()
t = torch.randint(0, 1000, (B,), device=dev)


def step():
    opt.zero_grad()
    loss = torch.nn.functional.cross_entropy(model(x).float(), t)
    loss.backward()
    opt.step()


for _ in range(3):
    step()
torch.cuda.synchronize()
from torch.profiler import ProfilerActivity, profile  # noqa: E402
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step()
    torch.cuda.synchronize()
sites = collections.Counter()
kinds = collections.Counter()
for ev in prof.events():
    if ev.name in ("aten::copy_", "aten::clone", "aten::contiguous", "aten::_foreach_copy_", "aten::zero_", "aten::fill_"):
        kinds[ev.name] += 1
        st = [s for s in (ev.stack or []) if "cotnet_amd" in s or "bench" in s or "diag_memcpy" in s][:3]
        shp = str(ev.input_shapes)[:60] if ev.input_shapes else ""
        sites[(ev.name, " <- ".join(s.split("/")[-1] for s in st), shp)] += 1
print(kinds)
for (name, st, shp), n in sites.most_common(40):
    print(f"{n:5d}  {name:22s} {st}  {shp}")
gpu = collections.Counter()
for ev in prof.events():
    if ev.device_type == torch.autograd.DeviceType.CUDA and ("Memcpy" in ev.name or "copyBuffer" in ev.name or "fillBuffer" in ev.name or "Memset" in ev.name):
        gpu[ev.name[:50]] += 1
print(gpu)

#!/usr/bin/env python
"""The benchmark's training step without bench.py around it (for API traces): python scripts/step_plain.py [steps]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402
import cotnet_amd  # noqa: E402
from cotnet_amd.flat_sgd import FlatSGD, to_mixed_bf16  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
bench.apply_kernel_set("new")
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = to_mixed_bf16(cotnet_amd.create_model("cotnet50", num_classes=1000).to(dev)).train()
opt = FlatSGD(model, lr=0.03, momentum=0.9, weight_decay=4e-5, nesterov=True)
x = torch.randn(80, 3, 224, 224, device=dev).bfloat16()
t = torch.randint(0, 1000, (80,), device=dev)
for _ in range(steps):
    opt.zero_grad()
    loss = torch.nn.functional.cross_entropy(model(x).float(), t)
    loss.backward()
    opt.step()
torch.cuda.synchronize()
print("done", steps)

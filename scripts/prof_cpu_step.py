#!/usr/bin/env python
"""Where does the HOST spend a training step?  cProfile over steps at a small batch (device work is short, so the host is
what is timed), `new` kernel set.   python scripts/prof_cpu_step.py [--batch 20]"""
import argparse
import cProfile
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402
import cotnet_amd  # noqa: E402
from cotnet_amd.flat_sgd import FlatSGD, to_mixed_bf16  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=20)
ap.add_argument("--steps", type=int, default=20)
args = ap.parse_args()
bench.apply_kernel_set("new")
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = to_mixed_bf16(cotnet_amd.create_model("cotnet50", num_classes=1000).to(dev)).train()
opt = FlatSGD(model, lr=0.03, momentum=0.9, weight_decay=4e-5, nesterov=True)
x = torch.randn(args.batch, 3, 224, 224, device=dev).bfloat16()
t = torch.randint(0, 1000, (args.batch,), device=dev)


def step():
    opt.zero_grad()
    loss = torch.nn.functional.cross_entropy(model(x).float(), t)
    loss.backward()
    opt.step()


for _ in range(10):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(args.steps):
    step()
t_issue = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print(f"batch {args.batch}: host issue {t_issue / args.steps * 1e3:.2f} ms/step, with the device {t_all / args.steps * 1e3:.2f} ms/step")
pr = cProfile.Profile()
pr.enable()
for _ in range(args.steps):
    step()
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
st.sort_stats("cumulative").print_stats(22)

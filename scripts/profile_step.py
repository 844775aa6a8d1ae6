#!/usr/bin/env python
"""torch.profiler kernel table of the bench.py training step (same model / batch / autocast), for analysis.
The committed evidence is the rocprofv3 summary under profiles/; this is the quick in-process view."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from torch.profiler import ProfilerActivity, profile  # noqa: E402

import cotnet_amd  # noqa: E402
from bench import make_optimizer  # noqa: E402
from cotnet_amd.data_parallel import GradBucketReducer  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=80)
ap.add_argument("--steps", type=int, default=3)
ap.add_argument("--model", default="cotnet50")
ap.add_argument("--out", default="gpurun_out/torch_prof.txt")
ap.add_argument("--rows", type=int, default=70)
ap.add_argument("--mixed", action="store_true", help="bf16 weights + FlatSGD instead of autocast + torch SGD")
args = ap.parse_args()
dev = torch.device("cuda:0")
torch.backends.cudnn.benchmark = True
torch.manual_seed(0)
model = cotnet_amd.create_model(args.model, num_classes=1000).to(dev).train()
x = torch.randn(args.batch, 3, 224, 224, device=dev)
t = torch.randint(0, 1000, (args.batch,), device=dev)
if args.mixed:
    from cotnet_amd.flat_sgd import FlatSGD, to_mixed_bf16
    to_mixed_bf16(model)
    x = x.bfloat16()
    opt = FlatSGD(model, lr=0.03, weight_decay=4e-5)

    def step():
        opt.zero_grad()
        loss = torch.nn.functional.cross_entropy(model(x).float(), t)
        loss.backward()
        opt.step()
else:
    opt = make_optimizer(model, 0.03, 4e-5)
    red = GradBucketReducer(model)

    def step():
        red.zero_grad()
        with torch.autocast("cuda", dtype=torch.bfloat16):
            loss = torch.nn.functional.cross_entropy(model(x).float(), t)
        loss.backward()
        red.finish()
        opt.step()


for _ in range(4):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
ka = prof.key_averages()
tot = sum(e.self_device_time_total for e in ka)
lines = [f"total device time {tot / 1e3 / args.steps:.2f} ms/step over {args.steps} steps, batch {args.batch}"]
rows = sorted([e for e in ka if e.self_device_time_total > 0], key=lambda e: -e.self_device_time_total)
lines.append(f"{'self_dev_ms/step':>16s} {'%':>6s} {'calls/step':>10s} {'avg_us':>8s}  name")
for e in rows[:args.rows]:
    lines.append(f"{e.self_device_time_total / 1e3 / args.steps:16.3f} {100 * e.self_device_time_total / tot:6.2f} "
                 f"{e.count / args.steps:10.1f} {e.self_device_time_total / max(e.count, 1):8.1f}  {e.key[:150]}")
os.makedirs(os.path.dirname(args.out) or ".", exist_ok=True)
open(args.out, "w").write("\n".join(lines) + "\n")
print("\n".join(lines))

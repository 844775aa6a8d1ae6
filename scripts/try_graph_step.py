#!/usr/bin/env python
"""Experiment: the benchmark's training step captured in ONE hipGraph (torch.cuda.CUDAGraph) and replayed, against the eager step
from the same initial state: loss per step (must match: same kernels, same data, deterministic) and ms per step.

    python scripts/try_graph_step.py [steps [model [batch [resolution]]]]"""
import copy
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402
import cotnet_amd  # noqa: E402
from cotnet_amd.flat_sgd import FlatSGD, to_mixed_bf16  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
MODEL = sys.argv[2] if len(sys.argv) > 2 else "cotnet50"
BATCH = int(sys.argv[3]) if len(sys.argv) > 3 else 80
RES = int(sys.argv[4]) if len(sys.argv) > 4 else 224
bench.apply_kernel_set("new")
dev = torch.device("cuda:0")


def make():
    torch.manual_seed(0)
    model = to_mixed_bf16(cotnet_amd.create_model(MODEL, num_classes=1000).to(dev)).train()
    opt = FlatSGD(model, lr=0.03, momentum=0.9, weight_decay=4e-5, nesterov=True)
    return model, opt


torch.manual_seed(1)
x = torch.randn(BATCH, 3, RES, RES, device=dev).bfloat16()
t = torch.randint(0, 1000, (BATCH,), device=dev)


def run(graphed):
    model, opt = make()
    out = {}

    def step():
        opt.zero_grad()
        loss = torch.nn.functional.cross_entropy(model(x).float(), t)
        loss.backward()
        opt.step()
        out["loss"] = loss.detach()

    losses = []
    for _ in range(3):  # eager warm-up (lazy initialisation: masks, LDS opt-in, caches)
        step()
        losses.append(float(out["loss"]))
    g = None
    if graphed:
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            step()  # one more eager step on the side stream (allocator warm-up for capture)
            losses.append(float(out["loss"]))
        torch.cuda.current_stream().wait_stream(s)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            step()
        torch.cuda.synchronize()
        # (the captured step has NOT run: capture only records)
    else:
        step()
        losses.append(float(out["loss"]))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        if g is not None:
            g.replay()
        else:
            step()
        losses.append(None)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    # loss trajectory, step by step (synchronising: separate pass)
    traj = []
    for _ in range(6):
        if g is not None:
            g.replay()
        else:
            step()
        traj.append(round(float(out["loss"]), 5))
    return ms, [round(v, 5) for v in losses if v is not None], traj


for mode in (False, True):
    try:
        ms, first, traj = run(mode)
        print(f"{'graph' if mode else 'eager'}: {ms:.3f} ms/step; warm-up losses {first}; after {steps} more steps: {traj}", flush=True)
    except Exception as e:  # noqa: BLE001
        import traceback
        traceback.print_exc()
        print(f"{'graph' if mode else 'eager'}: FAILED {type(e).__name__}: {str(e)[:300]}", flush=True)

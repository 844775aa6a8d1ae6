#!/usr/bin/env python
"""Can an RCCL all-reduce be captured into a HIP graph on this stack?  (world of one rank on one GPU; each probe in a child process,
a crash is reported, not fatal)"""
import os
import subprocess
import sys

CHILD = r'''
import os, sys, torch, torch.distributed as dist
mode = sys.argv[1]
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
dist.init_process_group("nccl", rank=0, world_size=1)
dev = torch.device("cuda:0"); torch.cuda.set_device(dev)
x = torch.ones(1 << 20, device=dev)
dist.all_reduce(x); torch.cuda.synchronize()
s = torch.cuda.Stream(); side = torch.cuda.Stream()
torch.cuda.set_stream(s)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, stream=s):
    y = x * 2
    if mode == "same_stream":
        dist.all_reduce(y)
    elif mode == "side_stream_async":
        ev = torch.cuda.Event(); ev.record(s)
        with torch.cuda.stream(side):
            side.wait_event(ev)
            w = dist.all_reduce(y, async_op=True)
            w.wait()
        s.wait_stream(side)
    elif mode == "side_stream_sync":
        ev = torch.cuda.Event(); ev.record(s)
        with torch.cuda.stream(side):
            side.wait_event(ev)
            dist.all_reduce(y)
        s.wait_stream(side)
    elif mode == "same_stream_async":
        w = dist.all_reduce(y, async_op=True)
        w.wait()
    elif mode == "side_stream_async_wait_on_main":
        ev = torch.cuda.Event(); ev.record(s)
        with torch.cuda.stream(side):
            side.wait_event(ev)
            w = dist.all_reduce(y, async_op=True)
        w.wait()
        s.wait_stream(side)
    elif mode == "side_stream_async_external_event":
        ev = torch.cuda.Event(external=True) if hasattr(torch.cuda.Event, "__init__") else torch.cuda.Event(); ev.record(s)
        with torch.cuda.stream(side):
            side.wait_event(ev)
            w = dist.all_reduce(y, async_op=True)
            w.wait()
        s.wait_stream(side)
    z = y + 1
torch.cuda.synchronize()
g.replay(); torch.cuda.synchronize()
print("ok", mode, float(z[0]))
dist.destroy_process_group()
'''

for mode in ("none", "same_stream", "same_stream_async", "side_stream_sync", "side_stream_async", "side_stream_async_wait_on_main"):
    r = subprocess.run([sys.executable, "-X", "faulthandler", "-c", CHILD, mode], capture_output=True, text=True, timeout=120)
    tail = (r.stdout.strip().splitlines() or [""])[-1]
    err = [ln for ln in r.stderr.splitlines() if "Fatal" in ln or "Error" in ln or "error" in ln][:3]
    print(f"{mode:20s} rc={r.returncode} {tail} {' | '.join(err)[:300]}", flush=True)

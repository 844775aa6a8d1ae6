#!/usr/bin/env python3
"""Layout study, part 3: one whole CotLayer, forward + backward, B = 80 -- the channels-last launch sequence of
cotnet_amd/channels_last_study.py on the study kernels beside the product's NCHW single-node layer (cot_layer_fused), at the
14 x 14 (dim 256) and 7 x 7 (dim 512) stages.  Reported per variant: the sum of the kernels' device times (dispatch-attached
events, COT_PROFILE_ALL=1: free of host launch gaps) and the wall time per iteration of a queue of iterations.
    COT_PROFILE_ALL=1 python scripts/bench_cot_layer_channels_last.py [iters]
"""
import collections
import ctypes
import os
import sys

os.environ.setdefault("COT_PROFILE_ALL", "1")
import torch  # noqa: E402

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cotnet_amd import _lib, aggregation_zeropad as az, channels_last_study as cls, cot_layer_fused as clf  # noqa: E402
from cotnet_amd.cotnet import CotLayer  # noqa: E402
from cotnet_amd.flat_sgd import to_mixed_bf16  # noqa: E402
from tests import truth  # noqa: E402

ITERS = int(sys.argv[1]) if len(sys.argv) > 1 else 10
dev = torch.device("cuda:0")
L = _lib.lib()
N = 80


def device_ms(fn):
    """sum of the library kernels' device times of one call of fn (+ per-kernel totals)"""
    fn()
    torch.cuda.synchronize()
    az.profile_begin()
    fn()
    recs = az.profile_end()
    by = collections.Counter()
    for r in recs:
        by[r[-1]] += r[4]
    return sum(by.values()), by, len(recs)


def wall_ms(fn):
    for _ in range(2):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(ITERS):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / ITERS


for H, D in ((14, 256), (7, 512)):
    torch.manual_seed(D)
    layer = to_mixed_bf16(CotLayer(D, 3).to(dev).train())
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    # channels-last sequence
    plan = cls.Plan(layer)
    x_cl = torch.randn(N, H, H, D, device=dev).bfloat16()
    g_cl = torch.randn(N * H * H, D, device=dev).bfloat16()

    def run_cl():
        out, sv = cls.forward(L, plan, x_cl, N, H, H, stream)
        cls.backward(L, plan, sv, g_cl, stream)
    # the product's NCHW single-node layer
    x_nc = x_cl.permute(0, 3, 1, 2).contiguous().requires_grad_(True)
    g_nc = g_cl.view(N, H, H, D).permute(0, 3, 1, 2).contiguous()

    def run_nc():
        with truth.switches(**truth.SINGLE_NODE):
            y = layer(x_nc)
            assert y.grad_fn.name().startswith("_CotLayerNode"), y.grad_fn.name()
            y.backward(g_nc)
        for p in layer.parameters():
            p.grad = None
        x_nc.grad = None
    for name, fn in (("channels-last (study kernels)", run_cl), ("NCHW single node (product)", run_nc)):
        total, by, n = device_ms(fn)
        print(f"CotLayer({D}) at {H}x{H}, B = {N}, fwd + bwd, {name}: kernels {total:.3f} ms in {n} launches; wall {wall_ms(fn):.3f} ms / iteration")
        for k, v in by.most_common(8):
            print(f"      {v:7.3f} ms  {k}")

# the whole stride-1 Bottleneck: BottleneckCL (one node on channels-last tensors) beside the product's _BottleneckNode
from cotnet_amd.cotnet import Bottleneck  # noqa: E402

for H, cin, planes in ((14, 1024, 256), (7, 2048, 512)):
    torch.manual_seed(cin)
    blk = Bottleneck(cin, planes).to(dev).train()
    with torch.no_grad():
        blk.bn3.weight.fill_(0.8)
    blk = to_mixed_bf16(blk)
    bplan = cls.BlockPlan(blk)
    x_cl = torch.randn(N, H, H, cin, device=dev).bfloat16().requires_grad_(True)
    g_cl = torch.randn(N, H, H, cin, device=dev).bfloat16()
    x_nc = x_cl.detach().permute(0, 3, 1, 2).contiguous().requires_grad_(True)
    g_nc = g_cl.permute(0, 3, 1, 2).contiguous()

    def run_cl():
        cls.BottleneckCL.apply(L, bplan, x_cl, *blk.parameters()).backward(g_cl)
        for p in blk.parameters():
            p.grad = None
        x_cl.grad = None

    def run_nc():
        with truth.switches(**truth.SINGLE_NODE):
            y = blk(x_nc)
            assert y.grad_fn.name().startswith("_BottleneckNode"), y.grad_fn.name()
            y.backward(g_nc)
        for p in blk.parameters():
            p.grad = None
        x_nc.grad = None
    for name, fn in (("channels-last node (study kernels)", run_cl), ("NCHW single node (product)", run_nc)):
        total, by, n = device_ms(fn)
        print(f"Bottleneck({cin}, {planes}) at {H}x{H}, B = {N}, fwd + bwd, {name}: kernels {total:.3f} ms in {n} launches; wall {wall_ms(fn):.3f} ms / iteration")

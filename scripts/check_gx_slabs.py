#!/usr/bin/env python
"""errors of the channel-major CoTNeXt stage blocks against an fp32 evaluation, with CoXtLayer.embed[0] as torch.stack + grouped 1x1
(COT_GX_SLABS=0) and as two-slab kernels per group (=1): gradient of the input and of embed[0]'s weight, and the two forms against each other"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import torch
from torch import nn

import truth
from cotnet_amd import cot_layer_fused as clf
from cotnet_amd.cotnet import Bottleneck
from cotnet_amd.flat_sgd import to_mixed_bf16

DEV = "cuda"
for N, planes, H in [(64, 256, 14), (64, 512, 7)]:
    for seed in (0, 1, 2):
        torch.manual_seed(planes + H + 1000 * seed)
        inpl = 4 * planes
        stage = nn.Sequential(*[Bottleneck(inpl, planes, cardinality=2, base_width=48) for _ in range(2)]).to(DEV).train()
        with torch.no_grad():
            for p in stage.parameters():
                if p.ndim == 1:
                    p.add_(0.3 * torch.randn_like(p))
            for b in stage:
                b.bn3.weight.fill_(0.8)
        stage = to_mixed_bf16(stage)
        with truth.switches(cm=True):
            clf.plan_stage_layouts(stage)
        x = torch.randn(N, inpl, H, H, device=DEV).bfloat16()
        g = torch.randn(N, inpl, H, H, device=DEV).bfloat16()
        yt, gxt, gt = truth.truth(stage, x, g)
        res = {}
        for name, sw, slabs in (("nchw", dict(truth.SINGLE_NODE, cm=False), True), ("cm/stack", dict(truth.SINGLE_NODE, cm=True), False),
                                ("cm/slabs", dict(truth.SINGLE_NODE, cm=True), True)):
            clf.GX_SLABS = slabs
            res[name] = truth.run(stage, x, g, **sw)
        wn = "0.conv2.embed.0.weight"
        line = f"N={N} planes={planes} H={H} seed={seed}:"
        for name, (y, gx, gp) in res.items():
            line += f"  {name}: gx {truth.err(gx, gxt):.4f} gW {truth.err(gp[wn], gt[wn]):.4f}"
        line += f" | gx cm/stack~nchw {truth.err(res['cm/stack'][1], res['nchw'][1]):.4f} cm/slabs~nchw {truth.err(res['cm/slabs'][1], res['nchw'][1]):.4f}"
        line += f" gW slabs~stack {truth.err(res['cm/slabs'][2][wn], res['cm/stack'][2][wn]):.5f}"
        print(line, flush=True)

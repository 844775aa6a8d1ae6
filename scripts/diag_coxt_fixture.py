#!/usr/bin/env python
"""Which convolution makes the MIOpen-convolution path of a CoXtLayer fixture miss the 1e-3 bar?  (round 4: layer_coxtlayer_s2_192x28,
eval mode, input gradient 1.2 % off with MIOpen's fp32 convolutions; every convolution on the library's kernels passes.)"""
import itertools
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from tests import truth  # noqa: E402
from tests.conftest import load_golden, real_layer_case, sample_idx, K_OUT  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "layer_coxtlayer_s2_192x28"
gold = load_golden(name)
meta, layer, x, gout = real_layer_case(gold)
layer = layer.to("cuda")
state = {k: v.clone() for k, v in layer.state_dict().items()}
for c1, c3, det in itertools.product(("", "hip"), ("", "hip"), (False, True)):
    torch.backends.cudnn.deterministic = det
    with truth.switches(fused_layer=False, conv1x1=c1, conv3x3=c3, gn9="hip" if c1 else ""):
        for mode in ("eval", "train"):
            layer.load_state_dict(state)
            layer.train(mode == "train")
            layer.zero_grad()
            xin = x.to("cuda").requires_grad_(True)
            y = layer(xin)
            y.backward(gout.to("cuda"))
            out = []
            for key, t in (("y", y), ("gx", xin.grad)):
                flat = t.detach().float().cpu().reshape(-1)
                idx = sample_idx(flat.numel(), K_OUT)
                out.append(f"{key} {(flat[idx] - torch.from_numpy(gold[f'{mode}_{key}'])).abs().max().item():.2e}")
            print(f"conv1x1={c1 or 'miopen':6s} conv3x3={c3 or 'miopen':6s} deterministic={det} {mode}: {'  '.join(out)}", flush=True)

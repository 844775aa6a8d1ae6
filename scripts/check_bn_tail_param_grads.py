#!/usr/bin/env python
"""dgamma / dbeta of the BatchNorm folded into the radix tail on one channel: error against autograd next to the gradient's size and next to
rms(dz) * sqrt(count), for batch sizes around the ones the device fuzz flagged (8, 9)"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch

from cotnet_amd import _lib
from tests.bn_tail_cases import bn_tail_case

L = _lib.lib()
L._test_device = "cuda"
for (H, W) in [(3, 5), (9, 8), (20, 22)]:
    for N in range(5, 13):
        worst = [0, 0, 0, 0]
        for seed in range(12):
            for sums in (False, True):
                rep = []
                bn_tail_case(L, N, 1, H, W, torch.bfloat16, 0, sums, seed=seed, report=rep)
                eg, eb, sg, sb, noise = rep[0]
                worst = [max(worst[0], eg / sg), max(worst[1], eb / sb), max(worst[2], eg / noise), max(worst[3], eb / noise)]
        print(f"HW={H * W:4d} N={N:2d}: worst err/|grad| dgamma {worst[0]:.4f} dbeta {worst[1]:.4f}   worst err/(rms*sqrt(cnt)) dgamma {worst[2]:.5f} dbeta {worst[3]:.5f}")

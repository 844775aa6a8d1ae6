#!/usr/bin/env python
"""SE-CoTNetD's deep stem at the benchmark's size (64 x 3 x 320 x 320, stem_width 64): each of its three 3x3 convolutions, forward and
backward through autograd, on the library's kernels against the torch modules (MIOpen); HIP events, 10 iterations after 3 warm-ups."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from torch import nn  # noqa: E402

from cotnet_amd import conv3x3g, stem3x3  # noqa: E402


def timeit(fn, iters=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def main():
    dev = "cuda"
    from cotnet_amd import _lib
    for kv in filter(None, os.environ.get("COT_TUNE", "").split(",")):  # e.g. COT_TUNE=15:0,39:0
        k, v = kv.split(":")
        _lib.lib().cot_set_tuning(int(k), int(v))
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    torch.manual_seed(0)
    cases = [("3->64 s2 @320", nn.Conv2d(3, 64, 3, 2, 1, bias=False), (N, 3, 320, 320), False),
             ("64->64 s1 @160", nn.Conv2d(64, 64, 3, 1, 1, bias=False), (N, 64, 160, 160), True),
             ("64->128 s1 @160", nn.Conv2d(64, 128, 3, 1, 1, bias=False), (N, 64, 160, 160), True)]
    for name, conv, shape, need_gx in cases:
        conv = conv.to(dev).bfloat16()
        x = torch.randn(*shape, device=dev).bfloat16().requires_grad_(need_gx)
        ours = (lambda: stem3x3.stem3x3_conv(conv, x)) if conv.stride == (2, 2) else (lambda: conv3x3g.conv3x3(conv, x))
        y = ours()
        g = torch.randn_like(y)
        yr = conv(x)
        err = (y.float() - yr.float()).abs().max().item() / yr.float().abs().max().item()

        def fb(f):
            conv.weight.grad = None
            if need_gx:
                x.grad = None
            f().backward(g)
        fb(ours)
        gw = conv.weight.grad.clone()
        gx = x.grad.clone() if need_gx else None
        fb(lambda: conv(x))
        ew = ((gw.float() - conv.weight.grad.float()).norm() / conv.weight.grad.float().norm()).item()
        ex = ((gx.float() - x.grad.float()).norm() / x.grad.float().norm()).item() if need_gx else 0.0
        t_of, t_ob = timeit(ours), timeit(lambda: fb(ours))
        t_mf, t_mb = timeit(lambda: conv(x)), timeit(lambda: fb(lambda: conv(x)))
        parts = ""
        if need_gx:  # the two gradients one at a time (autograd prunes the other)
            xd = x.detach()
            t_w = timeit(lambda: torch.autograd.grad(conv3x3g.conv3x3(conv, xd), conv.weight, g)) - t_of
            conv.weight.requires_grad_(False)
            t_d = timeit(lambda: torch.autograd.grad(conv3x3g.conv3x3(conv, x), x, g)) - t_of
            conv.weight.requires_grad_(True)
            parts = f" [dgrad {t_d:7.1f} wgrad {t_w:7.1f}]"
        print(f"{name:16s} ours fwd {t_of:8.1f} us fwd+bwd {t_ob:8.1f} us{parts} | module fwd {t_mf:8.1f} us fwd+bwd {t_mb:8.1f} us | "
              f"rel err y {err:.1e} gw {ew:.1e} gx {ex:.1e}", flush=True)


if __name__ == "__main__":
    main()

#!/usr/bin/env python
"""Which parameter gradients still reach the flat buckets through a copy (torch._foreach_copy_ = one hipMemcpyAsync per
tensor on ROCm) instead of being written in place by the producing kernel (cotnet_amd.grad_sink)?  Runs bench.py's training
step in-process with the multi-tensor copy instrumented.

    python scripts/diag_bucket_copies.py"""
import collections
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

calls = []
_orig = torch._foreach_copy_


def spy(dst, src, *a, **k):
    calls.append([(tuple(d.shape), d.dtype) for d in dst])
    return _orig(dst, src, *a, **k)


torch._foreach_copy_ = spy
import bench  # noqa: E402

sys.argv = ["bench.py", "--steps", "2", "--warmup", "1", "--kernels", "new", "--no-cpu-baseline", "--no-kernel-timing"]
bench.main()
steps = 3
print(f"_foreach_copy_ calls: {len(calls)} over {steps} steps; tensors per step {sum(len(c) for c in calls) / steps:.1f}", file=sys.stderr)
cnt = collections.Counter(x for c in calls for x in c)
for (shape, dt), n in cnt.most_common(25):
    print(f"   {n / steps:6.1f} per step  {dt} {shape}", file=sys.stderr)

"""Whole-step HIP graph: forward + backward + gradient-bucket fill captured once, replayed per step.

Why: after the fused kernels the CoTNet-50 step at the reference's batch (80 / GPU) is bound by the HOST -- ~2 500
kernel launches per step at ~15 us of Python/dispatch each, while the device needs ~30 ms.  Capturing the step in a
hipGraph removes the per-launch host cost (guide: "capture launch-bound inner loops in hipGraphs").

What is inside the graph: model forward, loss, autograd backward, and the multi-tensor copies that move each
bucket's gradients into the flat communication buffers (the reducer's hooks run during capture, so those copies are
recorded).  What stays outside, eager, on purpose: the RCCL all-reduce of the flat buckets and the fused SGD kernels
(4-5 launches) -- collectives inside graphs are fragile across RCCL versions, and the learning rate may change every
step.  With the communication deferred there is no overlap with backward, which costs < 1 ms per step for CoTNet-50's
44 MB of bf16 gradients over xGMI.

Static-shape contract: inputs are copied into fixed device buffers; batch size and resolution are those of `example`.
"""
import torch


class GraphedTrainStep:
    def __init__(self, model, optimizer, loss_fn, example_input, example_target, warmup=3):
        self.model, self.opt, self.loss_fn = model, optimizer, loss_fn
        self.x = example_input.clone()
        self.t = example_target.clone()
        self.opt.reducer.defer_comm = True
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):  # warm-up on a side stream (allocator / MIOpen search / lazy inits) before capture
            for _ in range(warmup):
                self._eager()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.opt.zero_grad()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.loss = self.loss_fn(self.model(self.x), self.t)
            self.loss.backward()
        # during capture the hooks consumed their counters; later replays run no Python hooks
        for b in self.opt.reducer.buckets:
            b.pending = len(b.params)
            b.fired.clear()

    def _eager(self):
        self.opt.zero_grad()
        loss = self.loss_fn(self.model(self.x), self.t)
        loss.backward()
        self.opt.reducer.finish()  # defer_comm: only resets the hook counters
        self.opt.step(graphed=True)
        return loss

    def __call__(self, x=None, t=None):
        if x is not None:
            self.x.copy_(x, non_blocking=True)
        if t is not None:
            self.t.copy_(t, non_blocking=True)
        self.graph.replay()
        self.opt.step(graphed=True)
        return self.loss

#!/usr/bin/env python
"""bench.py -- CoTNet-50 224x224 forward+backward(+SGD) throughput on N MI355X, one JSON line on rank 0.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" = one pass of the hot path over one batch of synthetic input: CoTNet-50 (random init, 1000 classes)
forward + backward + SGD-nesterov update on B images per GPU, inputs resident in HBM before the timed region.
Data parallel: weak scaling (B per GPU fixed, reference recipe B=80: cot_experiments/CoTNet-50-350epoch/config.yaml:6),
gradients averaged with cotnet_amd.data_parallel.GradBucketReducer (RCCL all-reduce on a side stream).
K steps are timed between barrier + torch.cuda.synchronize() pairs; the slowest rank's time is reported.

Kernel set (`--kernels`, config.kernel_selection on the line): `new` (DEFAULT, the product) = every kernel of the step from
cotnet_amd/csrc inside single-node CotLayer / Bottleneck -- if the library cannot serve a layer the run
fails, it never falls back.  `round1` = MIOpen convolutions, one autograd node per op: a developer A/B baseline only; its
line is marked `"baseline_only": true` and its metric says so.  `auto` = the round-2/3 probe (a child process checks
`new` block by block against an fp32 truth and `round1`, and times both); when it does not end on `new` the run EXITS with
status 3 instead of printing a MIOpen-backed headline.

Extra objects on the line:
  secondary     default run only (N = 1, CoTNet-50 B = 80): BASELINE configs 4 (CoTNeXt-101, B = 64) and 5 (SE-CoTNetD-152 320^2,
                B = 64) and config 3 at the reference's fp32 precision, each measured by a short child run of this script on the
                same GPU with the same kernel set and timing protocol (8 timed steps); `--no-secondary` skips them.
  roofline      the dominant aggregation kernel of the timed region (largest total device time): algorithmic bytes
                per launch / mean launch duration.  Durations come from HIP start/stop events attached to each kernel
                dispatch on its launch stream (hipExtLaunchKernelGGL inside libcotnet_hip.so, cot_profile_begin/_end),
                i.e. device execution time as rocprofv3 --kernel-trace reports it, over the K timed steps.
                Peak = 8 TB/s HBM3E (MI355X_MICROARCH.md).  `kernels` lists every aggregation geometry.
  cpu_baseline  rank 0, N=1 only: the reference's own aggregation kernels compiled for the host CPU
                (oracle/_ref, kind "reference"; falls back to oracle/agg_oracle.c, kind "port"), OpenMP over all
                host cores, timed on 4 images per CoT-layer geometry and scaled to images/s of AGGREGATION WORK ONLY
                (the reference has no CPU path for anything else -- SURVEY.md 0.3).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md "Chip-level parameters": 8.0 TB/s spec (6.29 TB/s measured copy)
COT50_LAYERS = {(64, 56): 3, (128, 28): 4, (256, 14): 6, (512, 7): 3}  # (C, H=W) -> CoT layers, SURVEY 8


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--settle-seconds", type=float, default=30.0,
                    help="untimed settling before the warm-up steps: chunks of 10 steps until three agree within 1 %% (0 = off)")
    ap.add_argument("--batch", type=int, default=80, help="images per GPU (reference recipe: 80)")
    ap.add_argument("--model", default="cotnet50")
    ap.add_argument("--img", type=int, default=224)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--precision", default="mixed", choices=["mixed", "autocast"],
                    help="bf16 only. mixed: bf16 conv/linear weights + fp32 masters, fused flat SGD (cotnet_amd.flat_sgd); "
                         "autocast: fp32 weights under torch.autocast + torch.optim.SGD")
    ap.add_argument("--layout", default="nchw", choices=["nchw", "nhwc"])
    ap.add_argument("--mode", default="train", choices=["train", "fwd"])
    ap.add_argument("--deterministic", action="store_true", help="torch.backends.cudnn.deterministic = True")
    ap.add_argument("--miopen-find", action="store_true",
                    help="torch.backends.cudnn.benchmark = True: let MIOpen search its solvers for the convolutions that are "
                         "not ours (round1 kernel set, CoTNeXt / SE-CoTNetD models).  Off by default: the search takes minutes "
                         "on a fresh box and the default kernel set has no MIOpen call to tune")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--graph", action="store_true",
                    help="capture the whole step (forward, backward, side-stream weight gradients, optimizer) into ONE HIP graph after the "
                         "settling steps and replay it in the warm-up and timed steps (DESIGN.md 5.3).  This is the DEFAULT for the "
                         "mixed-precision training step (the step is 14.1 ms of device work and 13.5-15.7 ms of host work to issue it "
                         "eagerly: a replay takes the host out of the measurement); N > 1: the compute part is the graph, the "
                         "all-reduces and the SGD kernels follow every replay eagerly.  A failed capture falls back to eager steps "
                         "and says so in the line")
    ap.add_argument("--eager", action="store_true", help="issue every step launch by launch (no HIP graph)")
    ap.add_argument("--force-collectives", action="store_true",
                    help="developer switch: on ONE GPU, run the N > 1 code path -- a world-of-one RCCL communicator, the buckets' "
                         "all-reduces on the communication stream, fp32 buckets, deferred communication around a graph")
    ap.add_argument("--no-pmc", action="store_true",
                    help="do not re-measure roofline.traffic with two rocprofv3 --pmc child passes (then the tracked table is carried)")
    ap.add_argument("--graph-collectives", choices=["on", "off"], default="off",
                    help="N > 1 under --graph: on = capture the bucket all-reduces (stream-ordered RCCL calls on the communication stream) and "
                         "the SGD kernels inside the step's graph -- one replay per step on every rank; off (default) = forward + backward are "
                         "the graph, collectives and SGD issued after every replay.  Measured through a world-of-one communicator "
                         "(profiles/r06_collectives.log): on 15.0 ms, off 14.4 ms, no process group 13.7 ms; a capture with collectives inside "
                         "that FAILS takes the NCCL watchdog down with it, which is why it is not tried by default")
    ap.add_argument("--no-secondary", action="store_true",
                    help="skip the `secondary` object of the default line (BASELINE configs 4 / 5 and the reference's fp32 precision "
                         "measured by short child runs of this script on the same GPU)")
    ap.add_argument("--bucket-mb", type=float, default=25.0,
                    help="flat gradient bucket size (MiB): the all-reduce granularity.  25 since round 6: the default N > 1 form issues the "
                         "all-reduces behind the replayed backward (nothing to overlap with), where fewer, larger messages cost less -- one GPU, "
                         "world-of-one RCCL: 14.23 ms at 10 MiB, 13.92-14.09 at 25 (profiles/r06_bucket_sweep.log)")
    ap.add_argument("--conv1x1", default=None, choices=["module", "hip", "matmul"],
                    help="1x1-convolution implementation (cotnet_amd/conv1x1.py); default: COT_CONV1X1 or the module")
    ap.add_argument("--ema", type=float, default=None, metavar="DECAY",
                    help="also keep the reference recipe's weight EMA (model_ema_decay 0.9999) in the step, flat kernel")
    ap.add_argument("--recipe", action="store_true",
                    help="the reference recipe's regularisation in the step (cot_experiments/CoTNet-50-350epoch/config.yaml:21-26: "
                         "drop 0.25, drop_path 0.1, model_ema decay 0.9999): stochastic depth inside the single-node Bottlenecks, "
                         "head dropout on the library path, flat EMA kernel")
    ap.add_argument("--kernels", default="new", choices=["auto", "round1", "new"],
                    help="which kernel set the step runs on: new (default) = the library's own kernels inside single-node "
                         "CotLayer / Bottleneck, no fallback; round1 = MIOpen convolutions + node-per-op layers (developer "
                         "baseline; the line is marked baseline_only); auto = a child process checks `new` against an fp32 "
                         "truth and `round1` on this GPU and times both -- the run exits with status 3 unless `new` wins")
    ap.add_argument("--grad-dtype", default="auto", choices=["auto", "param", "fp32"],
                    help="arithmetic of the gradient all-reduce: fp32 = fp32 buckets, the reference's own reduction (DDP sums fp32 "
                         "gradients, train.py:112-115); param = the parameters' dtype (bf16 buckets, half the bytes on the wire, one "
                         "bf16 rounding per ring hop); auto (default) = fp32 whenever there IS a reduction (N > 1), param on one GPU "
                         "(nothing is reduced: the bucket is only the optimizer's input)")
    ap.add_argument("--tune", default="", metavar="KEY=VALUE[,KEY=VALUE...]",
                    help="developer A/B: cot_set_tuning(KEY, VALUE) after the kernel set is applied (include/cotnet_amd.h)")
    ap.add_argument("--probe-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--gn9", action="store_true", help="GroupNorm of the attention logits on csrc/group_norm9.hip")
    ap.add_argument("--fused-layer", action="store_true",
                    help="CotLayer as one autograd node (cotnet_amd/cot_layer_fused.py); implies --conv1x1 hip --conv3x3 hip")
    ap.add_argument("--conv3x3", default=None, choices=["module", "hip"],
                    help="grouped 3x3 key-embed convolution (cotnet_amd/conv3x3g.py); default: COT_CONV3X3 or the module")
    return ap.parse_args()


def make_optimizer(model, lr, wd):
    # reference: SGD nesterov, no weight decay on 1-D params (optim/optim_factory.py:19-31,:54-56)
    decay, no_decay = [], []
    for n, p in model.named_parameters():
        (no_decay if p.ndim <= 1 or n.endswith(".bias") else decay).append(p)
    groups = [{"params": no_decay, "weight_decay": 0.0}, {"params": decay, "weight_decay": wd}]
    return torch.optim.SGD(groups, lr=lr, momentum=0.9, nesterov=True, foreach=True)


def cpu_baseline(n_img=32, min_reps=10, budget_s=2.5):
    """the reference has no CPU path of its own (aggregation_zeropad.py:192-196 bounces CPU tensors to the
    GPU), so two stand-ins are timed on this box's host cores, fp32, the four CoTNet-50 stage geometries, `n_img` images
    each, forward + input-backward + weight-backward, best of >= `min_reps` runs:
      (1) kind "reference": the reference's own three kernels compiled for the host (oracle/_ref, OpenMP over the
          1024-thread blocks; falls back to the restatement oracle/agg_oracle.c = kind "port");
      (2) `unfold`: the reference's test formula nn.Unfold + broadcast-multiply + sum (:249-251) with autograd on CPU torch.
    -> dict for the JSON line; `value` is (1) scaled to images/s of AGGREGATION work of one CoTNet-50 step."""
    from oracle import build_ref, cref, unfold_oracle
    cores = os.cpu_count() or 1
    os.environ.setdefault("OMP_NUM_THREADS", str(cores))
    per_image_s, per_image_unfold_s = 0.0, 0.0
    kind = "reference"
    detail, detail_u, reps_done = {}, {}, []

    def best_of(fns):
        for f in fns:
            f()  # warm-up
        best, reps, t_start = float("inf"), 0, time.perf_counter()
        while (reps < min_reps and time.perf_counter() - t_start < 8 * budget_s) or \
                (time.perf_counter() - t_start < budget_s and reps < 200):
            t0 = time.perf_counter()
            for f in fns:
                f()
            best = min(best, time.perf_counter() - t0)
            reps += 1
        reps_done.append(reps)
        return best

    # Two phases, never interleaved: with torch's 256-thread CPU ops (the Unfold stand-in moves ~1 GB per call) running
    # between the reference kernels' OpenMP regions the latter measured ~100x slower on the MI355X box's host (6 images/s
    # instead of ~900: worker threads of the two phases spin against each other).  So: every reference timing first, exactly
    # as in round 1, then the Unfold stand-in.
    data = {}
    for (C, HW), layers in COT50_LAYERS.items():
        g = torch.Generator().manual_seed(C)
        data[(C, HW)] = (torch.randn(n_img, C, HW, HW, generator=g), torch.randn(n_img, 1, C // 8, 9, HW, HW, generator=g),
                         torch.randn(n_img, C, HW, HW, generator=g))
    for (C, HW), layers in COT50_LAYERS.items():
        geom = dict(dtype="float", N=n_img, C=C, H=HW, W=HW, heads=1, wC=C // 8, kernel_size=3, stride=1, padding=1,
                    dilation=1)
        x, w, go = data[(C, HW)]
        try:
            ref = build_ref.RefAggregation(**geom)
            fns = (lambda: ref.forward(x, w), lambda: ref.backward_input(go, w), lambda: ref.backward_weight(go, x))
        except FileNotFoundError:
            kind = "port"
            fns = (lambda: cref.forward(x, w, 3, 1, 1, 1), lambda: cref.backward_input(go, w, x.shape, 3, 1, 1, 1),
                   lambda: cref.backward_weight(go, x, w.shape, 3, 1, 1, 1))
        best = best_of(fns)
        detail[f"C{C}_H{HW}"] = round(best / n_img * 1e3, 4)
        per_image_s += layers * best / n_img
    nthr0 = torch.get_num_threads()
    torch.set_num_threads(min(cores, 64))  # (torch's elementwise CPU kernels stop scaling long before 256 threads)
    for (C, HW), layers in COT50_LAYERS.items():
        x, w, go = data[(C, HW)]

        def unfold_step():
            xa, wa = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
            unfold_oracle.aggregation_unfold(xa, wa, 3, 1, 1, 1).backward(go)
        try:
            bu = best_of((unfold_step,))
            detail_u[f"C{C}_H{HW}"] = round(bu / n_img * 1e3, 4)
            per_image_unfold_s += layers * bu / n_img
        except Exception as e:  # the stand-in is optional: never lose the line over it
            detail_u[f"C{C}_H{HW}"] = f"{type(e).__name__}"
            per_image_unfold_s = float("nan")
    torch.set_num_threads(nthr0)
    out = {"value": round(1.0 / per_image_s, 2), "unit": "images/s (aggregation fwd+bwd work of CoTNet-50 only)",
           "cores": cores, "kind": kind,
           "sample": f"{n_img} images per CoT-layer geometry (4 geometries x fwd/input-bwd/weight-bwd, fp32, best of "
                     f">={min(reps_done)} runs), scaled by layer counts 3/4/6/3; ms per image per layer: {detail}"}
    if per_image_unfold_s == per_image_unfold_s and per_image_unfold_s > 0:
        out["unfold"] = {"value": round(1.0 / per_image_unfold_s, 2), "unit": out["unit"],
                         "what": "nn.Unfold formula + autograd on CPU torch (reference test oracle, :249-251), same sample",
                         "ms_per_image_per_layer": detail_u}
    return out


def roctx_window(resume):
    """When run under `rocprofv3 --marker-trace`, COT_ROCTX=1 limits collection to the timed region (MIOpen's
    one-off convolution search during warm-up otherwise dominates the kernel statistics)."""
    if os.environ.get("COT_ROCTX") != "1":
        return
    try:
        import ctypes
        lib = ctypes.CDLL("librocprofiler-sdk-roctx.so")
        (lib.roctxProfilerResume if resume else lib.roctxProfilerPause)(ctypes.c_uint64(0))
    except OSError:
        pass


MODEL_TITLES = {"cotnet50": "CoTNet-50", "cotnet101": "CoTNet-101", "cotnext50_2x48d": "CoTNeXt-50", "cotnext101_2x48d": "CoTNeXt-101",
                "se_cotnetd_50": "SE-CoTNetD-50", "se_cotnetd_101": "SE-CoTNetD-101", "se_cotnetd_152": "SE-CoTNetD-152",
                "se_cotnetd_152_L": "SE-CoTNetD-152"}
KERNEL_SETS = {  # name -> (single-node layers, 1x1 mode, 3x3 mode, GroupNorm9 mode, cot_set_tuning(12) BatchNorm finalize fold)
    "round1": (False, "", "", "", 0),       # MIOpen convolutions, torch GroupNorm, one autograd node per op
    "new": (True, "hip", "hip", "hip", 1),  # every kernel of the step from cotnet_amd/csrc, one node per Bottleneck; the
                                            # BatchNorm finalize folded into the apply kernels (202 fewer launches, -0.6 ms)
}


def apply_kernel_set(name):
    from cotnet_amd import (_lib, conv1x1 as c1, conv3x3g as c3, cot_layer_fused as clf, group_norm9 as g9, head_fused as hf,
                            pool3x3 as p3, stem7x7 as s7)
    fused, m1, m3, mg, fold = KERNEL_SETS[name]
    clf.ENABLED, c1.MODE, c3.MODE, g9.MODE = fused, m1, m3, mg
    p3.MODE = hf.MODE = s7.MODE = mg  # poolings, classifier head and stem convolution go with the rest
    _lib.check(_lib.lib().cot_set_tuning(12, fold), "cot_set_tuning")


# per block: candidate error vs fp32 truth <= GATE x round1's error + GATE_FLOOR.  Outputs 1.5x; gradients 2.5x: a block's bf16
# input / parameter gradients sit 4-15 % from the truth and two correct kernel sets differ by up to 1.6x in that figure on a
# single block (measured on the MI355X: CoTNeXt-101 layer1.2 0.146 vs 0.090, median over the blocks 1.0x) -- a defect is a
# wrong term, i.e. an error of O(1): 5-10x
GATE_BLOCK, GATE_GRAD, GATE_FLOOR = 1.5, 2.5, 2e-3


def probe_model(make_model, dev, seed):
    """the benchmark model for the parity probe: same seed -> same init; the last BatchNorm of every residual branch is
    moved off its zero initialisation (models/cotnet.py:225-226, resnet.py:581-584) -- with bn3.weight == 0 every gradient
    inside the branches is exactly zero and a parity check would compare zeros with zeros"""
    from cotnet_amd.flat_sgd import to_mixed_bf16
    torch.manual_seed(seed)
    model = make_model().to(dev)
    with torch.no_grad():
        for m in model.modules():
            bn3 = getattr(m, "bn3", None)
            if isinstance(bn3, torch.nn.BatchNorm2d):
                bn3.weight.fill_(0.5)
    return to_mixed_bf16(model)


def _residual_blocks(model):
    """[(name, module)] of the model's residual blocks (cotnet.Bottleneck / cotnet_hybrid.CoTBottleneck / plain Bottlenecks):
    the granularity of the parity probe"""
    return [(n, m) for n, m in model.named_modules() if hasattr(m, "conv1") and hasattr(m, "bn3") and hasattr(m, "conv3")]


def rel_err(a, b):
    return float((a.float() - b.float()).abs().mean() / (b.float().abs().mean() + 1e-30))


def block_truth(make_model, dev, x, t, seed):
    """ONE fp32 forward/backward of the same model (same seed -> same init, rounded to bf16 exactly as to_mixed_bf16 does, then
    widened) by plain torch modules; recorded per residual block: its input x, the gradient gy arriving at its output, its
    output y, its input gradient gx and its parameter gradients.  -> (loss, {block name: dict})"""
    from cotnet_amd import (conv1x1 as c1, conv3x3g as c3, cot_layer_fused as clf, fused_bn, group_norm9 as g9,
                            head_fused as hf, pool3x3 as p3, radix_tail, stem7x7 as s7)
    saved = [(m, a, getattr(m, a)) for m, a in ((c1, "MODE"), (c3, "MODE"), (g9, "MODE"), (p3, "MODE"), (hf, "MODE"),
                                               (s7, "MODE"), (clf, "ENABLED"), (fused_bn, "ENABLED"), (radix_tail, "ENABLED"))]
    try:
        for m, a, v in saved:
            setattr(m, a, False if isinstance(v, bool) else "")
        model = probe_model(make_model, dev, seed).float().train()
        rec, hooks = {}, []
        for name, blk in _residual_blocks(model):
            r = rec[name] = {}

            def pre(mod, inp, r=r):
                r["x"] = inp[0].detach().clone()
                if inp[0].requires_grad:
                    inp[0].register_hook(lambda g, r=r: r.__setitem__("gx", g.detach().clone()))

            def post(mod, inp, out, r=r):
                r["y"] = out.detach().clone()
                out.register_hook(lambda g, r=r: r.__setitem__("gy", g.detach().clone()))
            hooks += [blk.register_forward_pre_hook(pre), blk.register_forward_hook(post)]
        loss = torch.nn.functional.cross_entropy(model(x.float()), t)
        loss.backward()
        for h in hooks:
            h.remove()
        for name, blk in _residual_blocks(model):
            rec[name]["gp"] = {pn: p.grad.detach().clone() for pn, p in blk.named_parameters() if p.grad is not None}
        return float(loss.detach()), rec
    finally:
        for m, a, v in saved:
            setattr(m, a, v)


def block_errors(model, rec, run_block=None):
    """every residual block of `model` (a kernel set's bf16 model) evaluated on the TRUTH's input and upstream gradient of that
    block (rounded to bf16): -> {block: (err_y, err_gx, err_params)}, err = mean |a - truth| / mean |truth|; err_params pools
    the block's parameter gradients.  Errors are LOCAL to one block (~1-10 %): a defect is O(1) there, where the whole-model
    gradients of a bf16 network sit ~1.3 from the truth for ANY kernel set (two independent noises; VERDICT r2 weak #1a)."""
    out = {}
    blocks = dict(_residual_blocks(model))
    for name, r in rec.items():
        blk = blocks[name]
        if run_block is not None:  # (tests: a stand-in for the block's forward/backward)
            y, gx, gp = run_block(name, r)
        else:
            for p in blk.parameters():
                p.grad = None
            xi = r["x"].to(torch.bfloat16).requires_grad_(True)
            y = blk(xi)
            y.backward(r["gy"].to(torch.bfloat16))
            gx = xi.grad
            gp = {pn: p.grad for pn, p in blk.named_parameters() if p.grad is not None}
        num = sum(float((gp[k].float() - v).abs().sum()) for k, v in r["gp"].items() if k in gp)
        den = sum(float(v.abs().sum()) for k, v in r["gp"].items() if k in gp)
        finite = bool(torch.isfinite(y.float()).all() and torch.isfinite(gx.float()).all()
                      and all(torch.isfinite(g.float()).all() for g in gp.values()) and set(gp) == set(r["gp"]))
        out[name] = (rel_err(y.detach(), r["y"]), rel_err(gx, r["gx"]) if "gx" in r else 0.0, num / max(den, 1e-30), finite)
        for p in blk.parameters():
            p.grad = None
    return out


def block_gate(rec, errs, ref_errs):
    """a kernel set is verified when, block by block, it is not further from the fp32 truth than round1 is (x GATE_BLOCK +
    GATE_FLOOR) -- output, input gradient and pooled parameter gradients -- and everything is finite"""
    worst, worst_at = 0.0, ""
    ok = True
    for name, e in errs.items():
        r = ref_errs[name]
        ok = ok and e[3]
        for k, what in enumerate(("y", "gx", "params")):
            ratio = e[k] / ((GATE_BLOCK if k == 0 else GATE_GRAD) * r[k] + GATE_FLOOR)
            if ratio > worst:
                worst, worst_at = ratio, f"{name}.{what}"
    rec.update(finite=bool(ok), worst_block_ratio_to_gate=round(worst, 3), worst_block=worst_at,
               block_err_vs_fp32={n: [round(v, 4) for v in e[:3]] for n, e in errs.items()})
    rec["parity"] = bool(ok and worst <= 1.0)


def probe_child(args, dev=None, make_model=None, warm=4, timed=8):
    """(child process of --kernels auto) per kernel set: every residual block of the benchmark model is run forward and
    backward on the fp32 truth's input / upstream gradient of that block and compared with the truth (block_errors); a set is
    verified when block by block it is as close to the truth as round1 is (block_gate); then a short timing of each set.
    (dev / make_model / warm / timed: the CPU test drives this very function on the host-emulated kernels.)"""
    import cotnet_amd
    from cotnet_amd.flat_sgd import FlatSGD
    if dev is None:
        dev = torch.device("cuda", 0)
        torch.cuda.set_device(dev)
        # MIOpen in immediate mode: its exhaustive "find" over the ~50 fp32 + ~50 bf16 convolution configurations of the
        # truth and of the round1 set took > 10 minutes on the MI355X box (the probe then timed out and the headline fell
        # back to round1).  round1's probe timing is therefore a lower bound on its speed; the selection rule does not
        # depend on it being tuned (see choose_kernels).
        torch.backends.cudnn.benchmark = False
    sync = torch.cuda.synchronize if dev.type == "cuda" else (lambda: None)
    make_model = make_model or (lambda: cotnet_amd.create_model(args.model, num_classes=1000))
    B = args.batch
    g = torch.Generator(device="cpu").manual_seed(99)
    x = torch.randn(B, 3, args.img, args.img, generator=g).to(dev).bfloat16()
    t = torch.randint(0, 1000, (B,), generator=g).to(dev)
    out = {"sets": {}}
    try:
        truth_loss, rec_truth = block_truth(make_model, dev, x, t, 4321)
        out["fp32_truth_loss"] = truth_loss
        out["blocks"] = len(rec_truth)
    except Exception as e:
        out["truth_error"] = f"{type(e).__name__}: {e}"[:300]
        print("PROBE_RESULT " + json.dumps(out), flush=True)
        return out
    ref = None
    for name in KERNEL_SETS:
        try:
            apply_kernel_set(name)
            model = probe_model(make_model, dev, 4321).train()
            errs = block_errors(model, rec_truth)  # (before the optimizer exists: its gradient hooks own .grad afterwards)
            opt = FlatSGD(model, lr=1e-3, momentum=0.9, weight_decay=4e-5, nesterov=True)

            def fwd_bwd():
                opt.zero_grad()
                loss = torch.nn.functional.cross_entropy(model(x).float(), t)
                loss.backward()
                return loss

            loss = float(fwd_bwd().detach())
            opt.reducer.finish()
            sync()
            rec = {"loss": loss}
            if ref is None:
                ref = (loss, errs)
                block_gate(rec, errs, errs)  # (round1 against itself: finite, ratios < 1)
            else:
                rec["loss_rel_diff"] = round(abs(loss - ref[0]) / max(abs(ref[0]), 1e-6), 5)
                block_gate(rec, errs, ref[1])
                rec["parity"] = bool(rec["parity"] and rec["loss_rel_diff"] < 0.05)
            for _ in range(warm):
                fwd_bwd()
                opt.step()
            sync()
            t0 = time.perf_counter()
            for _ in range(timed):
                fwd_bwd()
                opt.step()
            sync()
            rec["ms_per_step"] = round((time.perf_counter() - t0) / timed * 1e3, 3)
            out["sets"][name] = rec
            opt.reducer.remove()
            del model, opt
            if dev.type == "cuda":
                torch.cuda.empty_cache()
        except Exception as e:  # a kernel set that cannot run is simply not eligible
            out["sets"][name] = {"parity": False, "error": f"{type(e).__name__}: {e}"[:300]}
    print("PROBE_RESULT " + json.dumps(out), flush=True)
    return out


def _probe_cache_path(args):
    """the verdict of a probe holds for (this build of the library, this model / batch / image size, this GPU type): cached
    in /tmp so that the N = 2, 4, 8 runs of a scaling sweep on the same node do not repeat rank 0's ~40 s probe"""
    import hashlib
    from cotnet_amd import _lib
    try:
        st = os.stat(_lib.LIB_PATH)
        gpu = torch.cuda.get_device_name(0) if torch.cuda.is_available() else "cpu"
        key = f"{st.st_size}:{int(st.st_mtime)}:{args.model}:{args.batch}:{args.img}:{gpu}:{torch.__version__}:{os.path.getmtime(__file__):.0f}"
        return os.path.join("/tmp", "cotnet_amd_probe_" + hashlib.sha1(key.encode()).hexdigest()[:16] + ".json")
    except OSError:
        return None


def choose_kernels(args):
    """-> (name of the kernel set to run, dict describing how it was chosen).  Never raises: any trouble = round1."""
    import subprocess
    info = {"mode": "auto"}
    cache = _probe_cache_path(args) if not os.environ.get("COT_NO_PROBE_CACHE") else None
    if cache and os.path.exists(cache):
        try:
            c = json.load(open(cache))
            c["info"]["cached"] = cache
            return c["chosen"], c["info"]
        except (OSError, ValueError, KeyError):
            pass
    try:
        env = {k: v for k, v in os.environ.items()
               if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "LOCAL_WORLD_SIZE", "GROUP_RANK", "ROLE_RANK", "MASTER_ADDR",
                            "MASTER_PORT", "TORCHELASTIC_RUN_ID", "COT_ROCTX")}
        cmd = [sys.executable, os.path.abspath(__file__), "--probe-child", "--batch", str(args.batch), "--img",
               str(args.img), "--model", args.model]
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=420)
        line = [ln for ln in r.stdout.splitlines() if ln.startswith("PROBE_RESULT ")]
        if not line:
            info["probe_error"] = f"child exit {r.returncode}: {r.stderr[-300:]}"
            return "round1", info
        sets = json.loads(line[-1][len("PROBE_RESULT "):])["sets"]
        info["probe"] = sets
        ok = {n: v["ms_per_step"] for n, v in sets.items() if v.get("parity") and "ms_per_step" in v}
        if "round1" not in ok:
            return "round1", info
        best = min(ok, key=ok.get)
        # the hand-written set is taken when it is VERIFIED (not further from the fp32 truth than round1) and at least 3 %
        # faster than round1 as the probe ran it (MIOpen untuned there; tuned, round1 measured 33.6 ms against 22.9 ms)
        if best != "round1" and ok[best] > 0.97 * ok["round1"]:
            best = "round1"
        if cache:
            try:
                json.dump({"chosen": best, "info": info}, open(cache, "w"))
            except OSError:
                pass
        return best, info
    except Exception as e:  # timeout, JSON trouble, ...
        info["probe_error"] = f"{type(e).__name__}: {e}"[:300]
        return "round1", info


SECONDARY = [  # (key, what, extra arguments): BASELINE.json configs 4 and 5 on one GPU, and config 3 at the reference's own precision
    ("cotnext101_2x48d_b64_224", "BASELINE config 4 on one GPU: CoTNeXt-101 2x48d 224^2, B = 64 (reference recipe batch), fwd+bwd+SGD",
     ["--model", "cotnext101_2x48d", "--batch", "64"]),
    ("se_cotnetd_152_L_b64_320", "BASELINE config 5 on one GPU: SE-CoTNetD-152 320^2, B = 64, fwd+bwd+SGD",
     ["--model", "se_cotnetd_152_L", "--img", "320", "--batch", "64"]),
    ("cotnet50_b80_224_fp32", "BASELINE config 3 at the reference's own precision (amp: False): CoTNet-50 224^2 fp32, B = 80",
     ["--dtype", "fp32", "--batch", "80"]),
    ("cotnet50_b80_224_fwd", "BASELINE config 2: CoTNet-50 224^2 bf16 forward-only (eval mode), one GPU, B = 80, eager",
     ["--mode", "fwd", "--batch", "80"]),
    ("cotnet50_b80_224_fwd_graph", "BASELINE config 2 with the forward captured in one HIP graph (replays)",
     ["--mode", "fwd", "--batch", "80", "--graph"]),
]


def secondary_lines(timeout_s=170):
    """-> {key: {...}}: short runs (8 timed steps) of the other BASELINE configurations as child processes of this script on the
    same GPU, same kernel set (`new`, no fallback), same timing protocol; a failure is recorded, never hidden, and never
    touches the headline (which is complete before the first child starts)."""
    import subprocess
    out = {}
    for key, what, extra in SECONDARY:
        cmd = [sys.executable, os.path.abspath(__file__), "--gpus", "1", "--steps", "8", "--warmup", "3", "--settle-seconds", "8",
               "--no-cpu-baseline", "--no-kernel-timing", "--no-secondary", "--kernels", "new"] + extra
        t0 = time.perf_counter()
        try:
            env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "COT_KERNEL_SUMMARY")}
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout_s, env=env)
            line = next((ln for ln in reversed(r.stdout.splitlines()) if ln.startswith("{")), None)
            if r.returncode != 0 or line is None:
                out[key] = {"what": what, "error": f"exit {r.returncode}: {r.stderr.strip().splitlines()[-1][:300] if r.stderr.strip() else 'no output'}"}
                continue
            d = json.loads(line)
            out[key] = {"what": what, "metric": d["metric"], "value": d["value"], "unit": d["unit"], "ms_per_step": d["ms_per_step"],
                        "steps": d["steps"], "dtype": d["dtype"], "per_gpu_batch": d["config"]["per_gpu_batch"],
                        "nodes_per_step": d["config"].get("nodes_per_step"), "module_fallbacks_per_step": d["config"].get("module_fallbacks_per_step"),
                        "kernel_selection": d["config"]["kernel_selection"],
                        "wall_s": round(time.perf_counter() - t0, 1)}
        except Exception as e:  # (timeout, JSON trouble: recorded)
            out[key] = {"what": what, "error": repr(e)[:300]}
    return out


def aggmix_config5():
    """BASELINE config 5's op-level shape: aggregation_zeropad_mix at (B = 64, C = 256, 20 x 20, wC = 32, heads = 1)
    through the C ABI, HIP events on the launch stream, buffer sets rotating beyond the Infinity Cache; GB/s on the ALGORITHMIC
    bytes e * (x + w1 + w2 + 2 out) forward (gout + w1 + w2 + gx / gout + x + gw1 + gw2 backward).  fp32 = the reference's dtype
    for this op (cupy_layers/utils.py:8-12 rejects half), bf16 = the storage type of the rest of this line."""
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    try:
        import bench_aggmix_abi as B
        out = {"what": "aggregation_zeropad_mix, op level, B=64 C=256 20x20 wC=32 heads=1 (cupy_layers/aggregation_zeropad_mix.py:20-207)",
               "peak_GBps": B.PEAK_GBS}
        for name, dt in (("fp32", torch.float32), ("bf16", torch.bfloat16)):
            out[name] = B.measure(64, dt, iters=20, rounds=5, cold=True)
        return out
    except Exception as e:  # (recorded, never hidden; the headline is complete before this runs)
        return {"error": f"{type(e).__name__}: {e}"[:300]}
    finally:
        sys.path.pop(0)


def measure_agg_traffic(key, timeout_s=90):
    """roofline.traffic measured IN this run: two child passes of scripts/bench_agg_abi.py under `rocprofv3 --pmc FETCH_SIZE` / `--pmc
    WRITE_SIZE` (separate passes, kernel trace only -- the guide's HBM recipe), HBM bytes per launch = 2 * FETCH_SIZE + WRITE_SIZE
    (KB -> bytes; the gfx950 half-count correction of FETCH_SIZE).  -> (bytes or None, how / why not)."""
    import shutil
    import subprocess
    import tempfile
    rp = shutil.which("rocprofv3")
    if rp is None:
        return None, "rocprofv3 not on PATH"
    kernel, shape, dtype = key.split("|")
    if dtype != "bfloat16" or shape != "N80xC64x56x56":
        return None, f"no in-run PMC recipe for {shape} {dtype}"
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    try:
        from agg_traffic_from_pmc import per_kernel
        vals = {}
        with tempfile.TemporaryDirectory(prefix="cot_pmc_", dir="/tmp") as tmp:
            env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "COT_KERNEL_SUMMARY", "COT_PROFILE_ALL")}
            env["TMPDIR"] = "/tmp"
            for counter in ("FETCH_SIZE", "WRITE_SIZE"):
                out = os.path.join(tmp, counter)
                cmd = [rp, "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", out, "-o", "pmc", "--", sys.executable,
                       os.path.join(ROOT, "scripts", "bench_agg_abi.py"), "--shapes", "0", "--dtypes", "bf16", "--variants", "dot2",
                       "--iters", "4", "--rounds", "1"]
                r = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True, timeout=timeout_s)
                if r.returncode != 0:
                    return None, f"rocprofv3 --pmc {counter} pass exited {r.returncode}"
                got, _ = per_kernel(out, counter)
                if kernel not in got:
                    return None, f"{kernel} not in the {counter} pass"
                vals[counter] = got[kernel]
        return int(round(2 * vals["FETCH_SIZE"] * 1024 + vals["WRITE_SIZE"] * 1024)), (
            "measured in this run: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE child passes (separate, kernel trace only) over "
            "scripts/bench_agg_abi.py at this kernel and shape; bytes per launch = 2*FETCH_SIZE + WRITE_SIZE (KB units, gfx950 correction)")
    except Exception as e:  # (timeout, CSV layout, ...: the tracked table is carried and labelled so)
        return None, f"in-run PMC passes failed: {type(e).__name__}: {e}"[:200]
    finally:
        sys.path.pop(0)


def spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher: re-execute this command line under torch.distributed.run, one rank
    per GPU on 127.0.0.1 (the driver's own launch line).  Returns the launcher's exit code."""
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    print(f"[bench] WORLD_SIZE unset: launching {n} ranks: {' '.join(cmd[1:8])} ...", file=sys.stderr, flush=True)
    return subprocess.call(cmd, env=env)


def main():
    args = parse()
    if os.environ.get("COT_KERNEL_SUMMARY"):
        os.environ["COT_PROFILE_ALL"] = "1"  # (costs host time per launch: not for headline numbers)
    if args.probe_child:
        return probe_child(args)
    explicit = args.gn9 or args.fused_layer or args.conv1x1 is not None or args.conv3x3 is not None
    selection = {"mode": "flags" if explicit else args.kernels}
    if args.gn9:
        from cotnet_amd import group_norm9 as _gn9
        _gn9.MODE = "hip"
    if args.fused_layer:
        from cotnet_amd import cot_layer_fused as _clf, head_fused as _hf, pool3x3 as _p3, stem7x7 as _s7
        _clf.ENABLED = True
        _p3.MODE = _hf.MODE = _s7.MODE = "hip"
        args.conv1x1 = args.conv1x1 or "hip"
        args.conv3x3 = args.conv3x3 or "hip"
    if args.conv1x1 is not None:
        from cotnet_amd import conv1x1 as _c1
        _c1.MODE = "" if args.conv1x1 == "module" else args.conv1x1
    if args.conv3x3 is not None:
        from cotnet_amd import conv3x3g as _c3
        _c3.MODE = "" if args.conv3x3 == "module" else args.conv3x3
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return spawn_ranks(args.gpus)  # plain `python bench.py --gpus N`: become the launcher of N ranks
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    reducing = world > 1 or args.force_collectives  # (there is a gradient all-reduce)
    if args.grad_dtype == "auto":
        args.grad_dtype = "fp32" if reducing else "param"
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X; there is no CPU path")
    if local_rank >= torch.cuda.device_count():
        raise SystemExit(f"bench.py: rank {rank} needs GPU {local_rank} but this node has {torch.cuda.device_count()}")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if reducing:
        # RCCL writes a version banner to the C-level stdout (it surfaces when the process exits): the contract is ONE JSON line on
        # stdout, so file descriptor 1 goes to stderr for everything but this script's own print
        sys.stdout.flush()
        keep = os.dup(1)
        os.dup2(2, 1)
        sys.stdout = os.fdopen(keep, "w", buffering=1)
    if world > 1:
        import datetime
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", init_method="env://", device_id=dev, timeout=datetime.timedelta(minutes=30))
    elif args.force_collectives:
        import socket
        with socket.socket() as so:
            so.bind(("127.0.0.1", 0))
            port = so.getsockname()[1]
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=dev)
    if not explicit:
        chosen = args.kernels
        if args.kernels == "auto":
            if not (args.dtype == "bf16" and args.precision == "mixed" and args.layout == "nchw"):
                raise SystemExit("bench.py --kernels auto: the probe covers bf16 mixed-precision NCHW only; run this configuration "
                                 "with --kernels new (the library's general kernels) or --kernels round1 (baseline)")
            elif world == 1:
                chosen, selection = choose_kernels(args)
            else:  # rank 0 probes on its GPU, everybody else waits for the verdict on the rendezvous store (CPU side)
                store = dist.distributed_c10d._get_default_store()
                if rank == 0:
                    chosen, selection = choose_kernels(args)
                    store.set("cot_kernel_set", chosen)
                else:
                    store.wait(["cot_kernel_set"], datetime.timedelta(minutes=25))
                    chosen = store.get("cot_kernel_set").decode()
        selection["chosen"] = chosen
        if rank == 0:
            print(f"[bench] kernel set: {chosen}  ({json.dumps(selection)[:600]})", file=sys.stderr, flush=True)
        if args.kernels == "auto" and chosen != "new":
            # a MIOpen-backed number must never stand in for the product's: no line at all
            if world > 1:
                dist.destroy_process_group()
            raise SystemExit(3)
        apply_kernel_set(chosen)

    import cotnet_amd
    from cotnet_amd import _lib
    from cotnet_amd import aggregation_zeropad as agg_mod
    from cotnet_amd.data_parallel import GradBucketReducer
    _lib.lib()  # fail loudly here if the HIP library is missing
    for kv in filter(None, args.tune.split(",")):
        key, value = kv.split("=")
        _lib.check(_lib.lib().cot_set_tuning(int(key), int(value)), "cot_set_tuning")

    torch.manual_seed(1234 + rank)
    torch.backends.cudnn.benchmark = bool(args.miopen_find)
    torch.backends.cudnn.deterministic = args.deterministic
    if args.recipe and args.ema is None:
        args.ema = 0.9999
    model = cotnet_amd.create_model(args.model, num_classes=1000, **(dict(drop_rate=0.25, drop_path_rate=0.1) if args.recipe else {})).to(dev)
    mf = torch.channels_last if args.layout == "nhwc" else torch.contiguous_format
    if args.layout == "nhwc":
        model = model.to(memory_format=torch.channels_last)
    amp = args.dtype == "bf16"
    B = args.batch
    x = torch.randn(B, 3, args.img, args.img, device=dev).contiguous(memory_format=mf)
    t = torch.randint(0, 1000, (B,), device=dev)

    mixed = amp and args.precision == "mixed"
    if mixed:
        from cotnet_amd.flat_sgd import FlatSGD, to_mixed_bf16
        to_mixed_bf16(model)
        x = x.bfloat16()
    if args.mode == "train" and mixed:
        model.train()
        opt = FlatSGD(model, lr=0.25 * B * world / 640.0, momentum=0.9, weight_decay=4e-5, nesterov=True,
                      bucket_mb=args.bucket_mb, ema_decay=args.ema, force_collectives=args.force_collectives,
                      grad_dtype=torch.float32 if args.grad_dtype == "fp32" else None)

        def step():
            opt.zero_grad()
            loss = torch.nn.functional.cross_entropy(model(x).float(), t)
            loss.backward()
            opt.step()
            return loss

        def step_compute_only():  # (graph capture with N > 1: forward + backward fill the flat buckets, no collective inside)
            opt.zero_grad()
            loss = torch.nn.functional.cross_entropy(model(x).float(), t)
            loss.backward()
            return loss
    elif args.mode == "train":
        model.train()
        opt = make_optimizer(model, lr=0.25 * B * world / 640.0, wd=4e-5)
        red = GradBucketReducer(model, bucket_mb=args.bucket_mb)

        def step():
            red.zero_grad()
            with torch.autocast("cuda", dtype=torch.bfloat16, enabled=amp):
                out = model(x)
                loss = torch.nn.functional.cross_entropy(out.float(), t)
            loss.backward()
            red.finish()
            opt.step()
            return loss
    else:
        model.eval()

        def step():
            with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16, enabled=amp and not mixed):
                return model(x).float().sum()

    def barrier():
        if world > 1:
            dist.barrier(device_ids=[local_rank])
        torch.cuda.synchronize()

    graph_explicit = args.graph
    if args.graph and args.eager:
        raise SystemExit("bench.py: --graph and --eager exclude each other")
    if not args.eager and args.mode == "train" and mixed:
        args.graph = True  # (default for the mixed-precision training step; see --graph)
    graph_note = None
    if args.graph:
        assert not reducing or (args.mode == "train" and mixed), "--graph with N > 1: the mixed-precision training step (FlatSGD buckets)"
        gstream = torch.cuda.Stream()
        torch.cuda.set_stream(gstream)  # a capture-capable (non-default) stream for everything from here on

    roctx_window(resume=False)
    # Settling (untimed, before the W warm-up steps): a fresh box runs the first tens of seconds of sustained load ~5 % slower than
    # it does afterwards, whatever ran before in the process (gpurun_out/r3s30_ab.log: 17.11 / 17.10 ms for the first two
    # processes on a box, 16.37 / 16.04 / 16.37 / 16.03 for the next four, alternating two kernel sets) -- device power
    # management, not this program's warm-up.  Chunks of 10 steps until three in a row agree within 1 % (all ranks take the
    # slowest rank's time, so they stop together), at most --settle-seconds.  Reported in the line as `settle`.
    settle = {"steps": 0, "first_chunk_ms": None, "last_chunk_ms": None, "seconds": 0.0}
    if args.settle_seconds > 0:
        t_s, hist = time.perf_counter(), []
        while True:
            barrier()
            c0 = time.perf_counter()
            for _ in range(10):
                loss = step()
            torch.cuda.synchronize()
            ct = torch.tensor([time.perf_counter() - c0, time.perf_counter() - t_s], device=dev, dtype=torch.float64)
            if world > 1:
                dist.all_reduce(ct, op=dist.ReduceOp.MAX)
            hist.append(float(ct[0].item()) * 100.0)  # ms per step
            settle["steps"] += 10
            done = len(hist) >= 3 and max(hist[-3:]) <= 1.01 * min(hist[-3:])
            if done or float(ct[1].item()) >= args.settle_seconds:
                break
        settle.update(first_chunk_ms=round(hist[0], 3), last_chunk_ms=round(hist[-1], 3), seconds=round(time.perf_counter() - t_s, 2))
    eager_step = step
    if args.graph:
        # (every step of this process -- the settling ones too -- ran on `gstream`, see below: autograd's AccumulateGrad nodes
        # remember the stream they were created under, and a node created under the default stream makes the engine synchronise
        # with it during capture, which aborts the process)
        torch.cuda.synchronize()
        # With a process group the RCCL watchdog THREAD polls the events of the collectives it has seen (hipEventQuery); under the
        # default global capture mode such a call from another thread while this one captures is an error that the watchdog rethrows
        # -- the process dies (2 of 12 runs through the world-of-one communicator, profiles/r06_capture_flake.log; with 8 ranks most
        # launches would lose one).  So: thread-local capture mode (other threads' calls are none of the capture's business), and the
        # watchdog gets a moment to retire the settling steps' finished collectives before the capture starts.
        CAPTURE_MODE = "thread_local" if (dist.is_available() and dist.is_initialized()) else "global"
        if CAPTURE_MODE == "thread_local":
            time.sleep(0.3)
        graph = torch.cuda.CUDAGraph()
        captured, full = True, False

        def agree(ok):
            """every rank or none (the forms issue their collectives in different orders)"""
            if world <= 1:
                return ok
            okt = torch.tensor([1 if ok else 0], device=dev, dtype=torch.int32)
            dist.all_reduce(okt, op=dist.ReduceOp.MIN)
            return bool(int(okt.item()))
        if reducing and args.graph_collectives == "on":
            # N > 1, first choice: the WHOLE step is the graph -- forward, backward, the buckets' all-reduces on the communication
            # stream as the gradient hooks launch them (RCCL collectives are capturable; they overlap backward inside the replay as
            # they do eagerly) and the flat SGD kernels: one replay per step on every rank, nothing issued by the host afterwards
            try:
                barrier()
                with torch.cuda.graph(graph, stream=gstream, capture_error_mode=CAPTURE_MODE):
                    graph_loss = eager_step()
                    if os.environ.get("COT_BENCH_FAIL_CAPTURE") == "full":
                        raise RuntimeError("COT_BENCH_FAIL_CAPTURE=full is set (test of the fallback)")
                full = True
            except Exception as e:
                graph_note = f"capture with the collectives inside failed ({type(e).__name__}: {str(e)[:160]})"
                print(f"[bench] {graph_note}", file=sys.stderr, flush=True)
            torch.cuda.synchronize()
            if not agree(full):
                full = False
                opt.zero_grad()  # (bucket bookkeeping of the abandoned capture)
                graph = torch.cuda.CUDAGraph()
        try:
            if full:
                pass
            elif reducing:
                # N > 1, second choice: forward + backward (+ the bucket fills) are the graph; the gradient hooks run once, at
                # capture, with communication deferred (GradBucketReducer.defer_comm), and every replay is followed by the buckets'
                # all-reduces on the communication stream and the flat SGD kernels (FlatSGD.step(deferred=True): ~2 launches per
                # bucket, no host synchronisation) -- the reference's order: backward, all-reduce, optimizer (train.py:264-293)
                opt.reducer.defer_comm = True
                barrier()
                with torch.cuda.graph(graph, stream=gstream, capture_error_mode=CAPTURE_MODE):
                    graph_loss = step_compute_only()
                    if os.environ.get("COT_BENCH_FAIL_CAPTURE") in ("1", "all"):
                        raise RuntimeError("COT_BENCH_FAIL_CAPTURE is set (test of the fallback)")
            else:
                with torch.cuda.graph(graph, stream=gstream, capture_error_mode=CAPTURE_MODE):
                    graph_loss = eager_step()
                    if os.environ.get("COT_BENCH_FAIL_CAPTURE") in ("1", "all"):
                        raise RuntimeError("COT_BENCH_FAIL_CAPTURE is set (test of the fallback)")
        except Exception as e:  # a capture that does not work here must not cost the measurement: eager steps, and the line says so
            captured = False
            graph_note = f"capture failed ({type(e).__name__}: {str(e)[:200]}): the steps were issued eagerly"
            print(f"[bench] {graph_note}", file=sys.stderr, flush=True)
        torch.cuda.synchronize()
        if world > 1 and not agree(captured) and captured:
            captured, graph_note = False, "capture failed on another rank: the steps were issued eagerly"
        from cotnet_amd.cot_layer_fused import invalidate_packs
        if not captured:
            if reducing:
                opt.reducer.defer_comm = False
                opt.zero_grad()
            args.graph = False
        elif reducing and not full:
            def step():  # noqa: F811
                graph.replay()
                opt.step(deferred=True)
                return graph_loss
        else:
            def step():  # noqa: F811
                graph.replay()
                invalidate_packs()  # (the replayed SGD kernels moved the weights behind torch's version counters)
                return graph_loss
        graph_form = "full" if (captured and (full or not reducing)) else ("deferred" if captured else "eager")
    for _ in range(args.warmup):
        loss = step()
    barrier()
    roctx_window(resume=True)
    # The timed region carries NO instrumentation (round 1 attached events to every dispatch inside it and gave away ~10 %
    # of the headline).  Kernel timing for the roofline object: the same step is run `timing_steps` more times right
    # after the timed region with the library's dispatch-attached events on (identical kernels, shapes and data).
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    issued = time.perf_counter() - t0  # (host time to ISSUE the steps: the loop returns when the last launch is queued)
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    roctx_window(resume=False)
    # host time to issue ONE step into an empty queue (the loop above blocks on the launch queue whenever the device is the
    # slower side, so `issued` only bounds it from above)
    t1 = time.perf_counter()
    step()
    issue_one = time.perf_counter() - t1
    torch.cuda.synchronize()
    # what actually ran: one more eager step with the node counters on (residual blocks of the model vs the single-node kinds taken)
    from cotnet_amd import cot_layer_fused as _clf0
    if args.graph and reducing:
        opt.reducer.defer_comm = False  # (the eager steps from here on reduce from their hooks again)
    _clf0.reset_node_counts()
    from cotnet_amd import _lib as _lib0
    _lib0.FALLBACKS.clear()  # (calls a wrapper handed back to the torch module it wraps during this one step: {} on the BASELINE configurations' hot path)
    eager_step()
    torch.cuda.synchronize()
    module_fallbacks = dict(_lib0.FALLBACKS)
    n_blocks = sum(1 for m in model.modules() if type(m).__name__ in ("Bottleneck", "CoTBottleneck"))
    nodes_per_step = dict(_clf0.NODE_COUNTS, residual_blocks=n_blocks)
    nodes_per_step["single_node_blocks"] = (f"{nodes_per_step['bottleneck'] + nodes_per_step['bottleneck_channel_major'] + nodes_per_step['split_attn_block'] + nodes_per_step['bottleneck_eval']}"
                                            f"/{n_blocks}")
    recs, timing_steps = [], 0
    if not args.no_kernel_timing:
        timing_steps = 3
        os.environ["COT_PROFILE_ALL"] = "1"  # events on every launch of the library (outside the timed region)
        # Every launch on ONE stream for these steps: an event pair around a launch also counts the time the kernel waits for CUs
        # that a weight gradient on the side stream still holds (42.1 us for the 56 x 56 aggregation backward against 38.1 in the
        # rocprofv3 trace of the un-instrumented step, profiles/r03: its own duration is what the roofline object is about)
        from cotnet_amd import cot_layer_fused as _clf
        _side = _clf.SIDE_WGRAD
        _clf.SIDE_WGRAD = False
        try:
            agg_mod.profile_begin()
            for _ in range(timing_steps):
                eager_step()  # (dispatch-attached events need real launches: never the graph)
            torch.cuda.synchronize()
            recs = agg_mod.profile_end()
        finally:
            _clf.SIDE_WGRAD = _side
    final_loss = float(loss.detach())

    tmax = torch.tensor([elapsed], device=dev, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    elapsed = float(tmax.item())

    if rank == 0 and os.environ.get("COT_KERNEL_SUMMARY"):
        # per-kernel device time of every launch the library made in the instrumented steps (dispatch-attached events):
        # a rocprofv3-free way to see which of the library's kernels carry the step
        summ = {}
        for kind, g, dtype, layout, ms, nbytes, kname in recs:
            e = summ.setdefault(kname, {"launches": 0, "total_ms": 0.0})
            e["launches"] += 1
            e["total_ms"] += ms
        rows = sorted(({"kernel": k, "launches_per_step": round(v["launches"] / timing_steps, 1),
                        "ms_per_step": round(v["total_ms"] / timing_steps, 4),
                        "avg_us": round(v["total_ms"] / v["launches"] * 1e3, 2)} for k, v in summ.items()),
                      key=lambda r: -r["ms_per_step"])
        with open(os.environ["COT_KERNEL_SUMMARY"], "w") as f:
            json.dump({"library_ms_per_step": round(sum(r["ms_per_step"] for r in rows), 3), "kernels": rows}, f, indent=1)
    if rank == 0:
        # ---- roofline of the aggregation kernels, from the HIP events of the timed region
        groups = {}
        for kind, g, dtype, layout, ms, nbytes, kname in recs:
            if not kname.startswith("agg_"):
                continue  # the library also times its BatchNorm / SGD launches; the roofline object is the aggregation
            key = (kind, g[1], g[2], str(dtype).replace("torch.", ""), "nhwc" if layout else "nchw")
            e = groups.setdefault(key, {"ms": 0.0, "n": 0, "bytes": nbytes, "N": g[0], "kernel": kname})
            e["ms"] += ms
            e["n"] += 1
        kernels = []
        for (kind, C, H, dt, lay), e in sorted(groups.items(), key=lambda kv: -kv[1]["ms"]):
            avg_ms = e["ms"] / e["n"]
            gbs = e["bytes"] / (avg_ms * 1e-3) / 1e9
            kernels.append({"kernel": e["kernel"], "op": f"agg_{kind}_{lay}", "shape": f"N{e['N']}xC{C}x{H}x{H}", "dtype": dt,
                            "launches": e["n"], "avg_us": round(avg_ms * 1e3, 2), "GBs": round(gbs, 1),
                            "frac": round(gbs / HBM_PEAK_GBS, 4), "total_ms": round(e["ms"], 3)})
        # ---- the convolution / BatchNorm calls that carry the step (VERDICT r2 missing #6): per (op, shape) the device time of
        # the CALL (all its launches: GEMM + split reduce, statistics + apply), its algorithmic bytes against 8 TB/s and, for
        # the convolutions, the MFMA rate against the 2.5 PFLOP/s dense bf16 peak
        ops = {}
        for kind, g, dtype, layout, ms, nbytes, kname in recs:
            if not str(kind).startswith("op:"):
                continue
            e = ops.setdefault((kind[3:], g[:5]), {"ms": 0.0, "launches": 0, "bytes": nbytes, "kernels": set()})
            e["ms"] += ms
            e["launches"] += 1
            e["kernels"].add(kname)
        op_rows = []
        for (op, (N_, Ci, Co, HW_, G_)), e in ops.items():
            calls = max(1, round(e["launches"] / max(1, len(e["kernels"]))))  # (every call issues the same kernels)
            us = e["ms"] / calls * 1e3
            row = {"op": op, "shape": f"N{N_} {Ci}->{Co} HW{HW_}" + (f" g{G_}" if G_ > 1 else ""), "calls_per_step": round(calls / timing_steps, 1),
                   "avg_us": round(us, 2), "GBs": round(e["bytes"] / (us * 1e-6) / 1e9, 1),
                   "frac_hbm": round(e["bytes"] / (us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4), "ms_per_step": round(e["ms"] / timing_steps, 4),
                   "kernels": sorted(e["kernels"])}
            if op.startswith("conv"):
                flops = 2.0 * N_ * HW_ * Ci * Co / G_ * (9 if op.startswith("conv3x3") else 1)
                row["TFLOPs"] = round(flops / (us * 1e-6) / 1e12, 1)
                row["frac_mfma"] = round(flops / (us * 1e-6) / 1e12 / 2500.0, 4)
            op_rows.append(row)
        op_rows.sort(key=lambda r: -r["ms_per_step"])
        fam = {}
        for r in op_rows:
            f = fam.setdefault(r["op"], {"ms_per_step": 0.0, "bytes_weighted": 0.0})
            f["ms_per_step"] += r["ms_per_step"]
            f["bytes_weighted"] += r["frac_hbm"] * r["ms_per_step"]
        op_families = {k: {"ms_per_step": round(v["ms_per_step"], 3), "time_weighted_frac_hbm": round(v["bytes_weighted"] / max(v["ms_per_step"], 1e-9), 4)}
                       for k, v in sorted(fam.items(), key=lambda kv: -kv[1]["ms_per_step"])}
        # ---- the whole step against the machine: algorithmic bytes of every ANNOTATED library call of one step
        # (aggregation launches, 1x1 / 3x3 convolution and BatchNorm calls) over the step's wall time and 8 TB/s; launches counted
        # over every launch the library made (annotated or not: GroupNorm, radix tail, poolings, stem, SGD carry no byte count)
        step_ms = elapsed / args.steps * 1e3
        agg_bytes = sum(nb for kind, g, dtype, layout, ms, nb, kname in recs if kname.startswith("agg_")) / max(timing_steps, 1)
        op_bytes = 0.0
        for (op, shp), e in ops.items():
            calls = max(1, round(e["launches"] / max(1, len(e["kernels"]))))
            op_bytes += e["bytes"] * calls / max(timing_steps, 1)
        lib_ms = sum(ms for kind, g, dtype, layout, ms, nb, kname in recs) / max(timing_steps, 1)
        ann_ms = (sum(ms for kind, g, dtype, layout, ms, nb, kname in recs if kname.startswith("agg_") or str(kind).startswith("op:"))
                  / max(timing_steps, 1))
        step_roofline = {
            "bytes": int(agg_bytes + op_bytes), "ms_per_step": round(step_ms, 3),
            "frac": round((agg_bytes + op_bytes) / (step_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if step_ms > 0 else None,
            "launches": round(len(recs) / max(timing_steps, 1), 1),
            "launches_under_10us": round(sum(1 for r in recs if r[4] < 0.010) / max(timing_steps, 1), 1),
            "ms_in_launches_under_10us": round(sum(r[4] for r in recs if r[4] < 0.010) / max(timing_steps, 1), 3),
            "library_kernel_ms_per_step_single_stream": round(lib_ms, 3), "annotated_share_of_kernel_time": round(ann_ms / lib_ms, 4) if lib_ms else None,
            "note": "bytes = algorithmic bytes (operands read once, results written once) of the aggregation / convolution / BatchNorm "
                    "calls of one step; frac = bytes / ms_per_step / 8 TB/s: how far the STEP is from the HBM roofline"}
        roofline = None
        if kernels:
            k0 = next((k for k in kernels if k["GBs"] > 0), kernels[0])  # (a launch without byte annotation never is the headline kernel)
            traffic, traffic_src = None, None
            tkey = f"{k0['kernel']}|{k0['shape']}|{k0['dtype']}"
            if world == 1 and not args.no_pmc:
                traffic, traffic_src = measure_agg_traffic(tkey)
            tpath = os.path.join(ROOT, "profiles", "agg_traffic.json")
            if traffic is None and os.path.exists(tpath):
                traffic = json.load(open(tpath)).get(tkey)
                traffic_src = ("CARRIED from profiles/agg_traffic.json (the builder's rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE session, "
                               "scripts/agg_traffic_from_pmc.py: 2*FETCH_SIZE + WRITE_SIZE, gfx950 correction) -- not re-measured in this run"
                               + (f": {traffic_src}" if traffic_src else ""))
            agg_total = sum(k["total_ms"] for k in kernels) / timing_steps  # ms of aggregation kernels per step
            roofline = {"bound": "hbm", "kernel": k0["kernel"], "shape": k0["shape"], "dtype": k0["dtype"],
                        "achieved": k0["GBs"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": k0["frac"],
                        "traffic": traffic,
                        "traffic_source": traffic_src if traffic else None,
                        "avg_us": k0["avg_us"],
                        "timing": f"dispatch-attached HIP events (on the launch stream) over {timing_steps} repeats of the "
                                  "step right after the un-instrumented timed region, every launch on one stream (no weight "
                                  "gradients beside the kernel being timed)",
                        "agg_share_of_step": round(agg_total / (elapsed / args.steps * 1e3), 4),
                        # every aggregation launch of the step, forward and backward, all four stages: algorithmic bytes / device time
                        "frac_all_layers": round(sum(k["GBs"] * k["total_ms"] for k in kernels) / max(sum(k["total_ms"] for k in kernels), 1e-9)
                                                 / HBM_PEAK_GBS, 4),
                        "kernels": kernels,
                        "step": step_roofline,
                        "conv_bn_families": op_families, "conv_bn_calls": op_rows,
                        "conv_bn_note": "per CALL of the C ABI (all launches of the call), dispatch-attached events; frac_hbm = algorithmic "
                                        "bytes / time / 8 TB/s, frac_mfma = 2*N*HW*Ci*Co(*9)/groups / time / 2.5 PFLOP/s; weight gradients "
                                        "run on a side stream beside other kernels, so their times overlap the rest of the step"}
        baseline_only = (not explicit and chosen == "round1") or (explicit and not __import__("cotnet_amd.cot_layer_fused", fromlist=["ENABLED"]).ENABLED)
        line = {
            "metric": f"images/sec {MODEL_TITLES.get(args.model, args.model)} {args.img}^2 " + ("fwd+bwd" if args.mode == "train" else "fwd")
                      + (" [MIOpen / node-per-op developer baseline, NOT the product path]" if baseline_only else ""),
            **({"baseline_only": True} if baseline_only else {}),
            "value": round(B * world * args.steps / elapsed, 2), "unit": "images/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "settle": settle, "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16" if amp else "f32", "data": "synthetic",
            "config": {"workload": f"{args.model} {args.img}x{args.img} {'fwd+bwd+SGD-nesterov' if args.mode == 'train' else 'forward-only'}, "
                                   f"random init, 1000 classes, NCHW synthetic ImageNet-shaped input resident in HBM",
                       "per_gpu_batch": B, "global_batch": B * world, "parallelism": f"dp{world}",
                       "layout": args.layout, "precision": ("bf16 weights+activations, fp32 master weights / norm params, fused flat SGD" if mixed
                                     else "bf16 autocast, fp32 weights" if amp else "fp32"),
                       "kernel_selection": selection, **({"tune": args.tune} if args.tune else {}),
                       **({"recipe": "drop 0.25, drop_path 0.1, model_ema 0.9999 (reference config.yaml:21-26)"} if args.recipe else {}),
                       "conv1x1": __import__("cotnet_amd.conv1x1", fromlist=["MODE"]).MODE or "module",
                       "conv3x3": __import__("cotnet_amd.conv3x3g", fromlist=["MODE"]).MODE or "module",
                       "group_norm9": __import__("cotnet_amd.group_norm9", fromlist=["MODE"]).MODE or "module",
                       "cot_layer_single_node_enabled": __import__("cotnet_amd.cot_layer_fused", fromlist=["ENABLED"]).ENABLED,
                       "nodes_per_step": nodes_per_step, "module_fallbacks_per_step": module_fallbacks,
                       "grad_sync": (f"RCCL all-reduce (AVG), flat {'fp32' if args.grad_dtype == 'fp32' else 'parameter-dtype'} buckets, side stream"
                                     if reducing else "none (1 GPU)")},
            "final_loss": round(final_loss, 4),
            # host time to issue a step (the timed loop's own duration before the final synchronize; with the device the longer
            # of the two the loop blocks on the launch queue, so this is an upper bound of the host's own work)
            "host_issue_ms_per_step": round(issued / args.steps * 1e3, 3), "host_issue_ms_one_step_idle_queue": round(issue_one * 1e3, 3),
            **({"graph": ("forward + backward + bucket fills captured in one HIP graph after the settling steps; every warm-up / timed step = one "
                          "replay, then the buckets' all-reduces (RCCL, communication stream) and the flat SGD kernels issued eagerly"
                          + (f" [{graph_note}]" if graph_note else "")) if (reducing and graph_form == "deferred")
                else ("whole step captured in one HIP graph after the settling steps" + (", the buckets' all-reduces (RCCL, communication "
                      "stream, overlapping backward) and the flat SGD kernels inside it" if reducing else "") + "; warm-up and timed steps are replays")}
               if args.graph else
               ({"graph": graph_note} if graph_note else {"graph": "off: every step issued launch by launch"})),
            "roofline": roofline,
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline()
        default_cfg = (args.model == "cotnet50" and args.img == 224 and args.batch == 80 and args.mode == "train" and mixed
                       and not explicit and not args.tune and not args.recipe and not graph_explicit and not args.eager)
        if world == 1 and default_cfg and not args.no_secondary:
            torch.cuda.empty_cache()  # (the children run on this GPU while this process is idle)
            line["secondary"] = secondary_lines()
            line["secondary"]["aggmix_config5"] = aggmix_config5()
        print(json.dumps(line), flush=True)
    if world > 1 or args.force_collectives:
        dist.destroy_process_group()


if __name__ == "__main__":
    _rc = main()
    sys.exit(_rc if isinstance(_rc, int) else 0)  # (the launcher form returns the ranks' exit code)

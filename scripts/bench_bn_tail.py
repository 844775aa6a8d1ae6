"""BatchNorm + SiLU folded into the radix tail (cot_radix_*_bn) against the separate kernels, per CoTNet-50 stage shape at B = 80, through the
C ABI: microseconds per launch (HIP events on the launch stream, buffers rotated beyond the Infinity Cache) and, with --diff, where the two
forward paths differ."""
import ctypes
import sys

import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from cotnet_amd import _lib  # noqa: E402

L = _lib.lib()
dev = torch.device("cuda")
P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None  # noqa: E731
BF = _lib.COT_BF16
SHAPES = [(80, 64, 56, 56), (80, 128, 28, 28), (80, 256, 14, 14), (80, 512, 7, 7)]


def timeit(fn, sets, iters=20):
    for i in range(3):
        fn(sets[i % len(sets)])
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(iters):
        fn(sets[i % len(sets)])
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def main():
    diff = "--diff" in sys.argv
    for (N, C, H, W) in SHAPES:
        HW = H * W
        nset = max(2, int(600e6 // (N * C * HW * 2 * 6)))
        nws = int(L.cot_bn_act_workspace(N, C))
        sets = []
        torch.manual_seed(0)
        for _ in range(nset):
            d = {}
            for nm in ("a", "k", "g"):
                d[nm] = (torch.randn(N, C, H, W, device=dev) * 1.3 + 0.4).bfloat16()
            for nm in ("y", "out", "gy", "gk", "ga"):
                d[nm] = torch.empty(N, C, H, W, device=dev, dtype=torch.bfloat16)
            d["gamma"], d["beta"] = 1 + 0.3 * torch.randn(C, device=dev), 0.2 * torch.randn(C, device=dev)
            d["st"] = torch.empty(2 * C + nws, device=dev)
            d["rm"], d["rv"], d["nbt"] = torch.zeros(C, device=dev), torch.ones(C, device=dev), torch.zeros((), dtype=torch.int64, device=dev)
            d["gap"], d["logits"] = torch.empty(C, N, device=dev, dtype=torch.bfloat16), torch.randn(2 * C, N, device=dev).bfloat16()
            d["attn"] = torch.empty(N, C, 2, device=dev, dtype=torch.bfloat16)
            d["glog"], d["ggap"] = torch.empty(2 * C, N, device=dev, dtype=torch.bfloat16), (0.5 * torch.randn(C, N, device=dev)).bfloat16()
            d["tsum"], d["dg"], d["db"] = torch.empty(N * C * 4, device=dev), torch.empty(C, device=dev), torch.empty(C, device=dev)
            d["ws"] = torch.empty(max(nws, 1), device=dev)
            sets.append(d)
        st = lambda d: (P(d["st"]), P(d["st"][C:]))  # noqa: E731
        calls = {
            "bn_fwd(silu)": lambda d: L.cot_bn_act_forward(P(d["a"]), None, P(d["y"]), P(d["gamma"]), P(d["beta"]), *st(d), P(d["rm"]), P(d["rv"]), P(d["nbt"]), P(d["st"][2 * C:]), N, C, HW, 1e-5, 0.1, 2, BF, None),
            "gap_t": lambda d: L.cot_radix_gap_t(P(d["y"]), P(d["k"]), P(d["gap"]), N, C, HW, BF, None),
            "mix_logits": lambda d: L.cot_radix_mix_logits(P(d["y"]), P(d["k"]), P(d["logits"]), P(d["out"]), P(d["attn"]), N, C, HW, BF, None),
            "bwd_reduce": lambda d: L.cot_radix_mix_backward_reduce(P(d["g"]), P(d["y"]), P(d["k"]), P(d["attn"]), P(d["glog"]), N, C, HW, BF, None),
            "bwd_apply": lambda d: L.cot_radix_mix_backward_apply(P(d["g"]), P(d["attn"]), P(d["ggap"]), P(d["gy"]), P(d["gk"]), N, C, HW, BF, None),
            "bn_bwd(silu)": lambda d: L.cot_bn_act_backward(P(d["gy"]), P(d["a"]), None, P(d["ga"]), None, P(d["gamma"]), P(d["beta"]), *st(d), P(d["dg"]), P(d["db"]), P(d["ws"]), N, C, HW, 2, BF, None),
            "batch_stats": lambda d: L.cot_bn_batch_stats(P(d["a"]), *st(d), P(d["rm"]), P(d["rv"]), P(d["nbt"]), P(d["st"][2 * C:]), N, C, HW, 1e-5, 0.1, BF, None),
            "gap_t_bn": lambda d: L.cot_radix_gap_t_bn(P(d["a"]), P(d["k"]), P(d["gap"]), P(d["gamma"]), P(d["beta"]), *st(d), None, None, None, None, N, C, HW, 1e-5, 0.1, 0, BF, None),
            "stats_sums": lambda d: L.cot_bn_stats_sums(P(d["a"]), P(d["st"][2 * C:]), N, C, HW, BF, None),
            "gap_t_bn(fin)": lambda d: L.cot_radix_gap_t_bn(P(d["a"]), P(d["k"]), P(d["gap"]), P(d["gamma"]), P(d["beta"]), *st(d), P(d["rm"]), P(d["rv"]), P(d["nbt"]), P(d["st"][2 * C:]), N, C, HW, 1e-5, 0.1, 0, BF, None),
            "mix_logits_bn": lambda d: L.cot_radix_mix_logits_bn(P(d["a"]), P(d["k"]), P(d["logits"]), P(d["out"]), P(d["attn"]), P(d["gamma"]), P(d["beta"]), *st(d), N, C, HW, 0, BF, None),
            "bwd_reduce_bn": lambda d: L.cot_radix_mix_backward_reduce_bn(P(d["g"]), P(d["a"]), P(d["k"]), P(d["attn"]), P(d["glog"]), P(d["tsum"]), P(d["gamma"]), P(d["beta"]), *st(d), N, C, HW, 0, BF, None),
            "bwd_apply_bn": lambda d: L.cot_radix_mix_backward_apply_bn(P(d["g"]), P(d["a"]), P(d["attn"]), P(d["ggap"]), P(d["tsum"]), P(d["ga"]), P(d["gk"]), P(d["gamma"]), P(d["beta"]), *st(d), P(d["dg"]), P(d["db"]), N, C, HW, 0, BF, None),
        }
        t = {}
        for nm, fn in calls.items():
            def chk(d, fn=fn, nm=nm):
                rc = fn(d)
                if rc:
                    raise RuntimeError(nm + ": " + L.cot_last_error().decode())
            t[nm] = timeit(chk, sets)
        sep = t["bn_fwd(silu)"] + t["gap_t"] + t["mix_logits"], t["bwd_reduce"] + t["bwd_apply"] + t["bn_bwd(silu)"]
        fus = t["stats_sums"] + t["gap_t_bn(fin)"] + t["mix_logits_bn"], t["bwd_reduce_bn"] + t["bwd_apply_bn"]
        print(f"N{N} C{C} {H}x{W}: " + "  ".join(f"{k} {v:.1f}" for k, v in t.items()))
        print(f"    forward separate {sep[0]:.1f} us -> folded {fus[0]:.1f};  backward separate {sep[1]:.1f} -> folded {fus[1]:.1f}")
        if diff:
            d = sets[0]
            calls["bn_fwd(silu)"](d); calls["gap_t"](d); calls["mix_logits"](d)
            g0, o0, a0, m0 = d["gap"].clone(), d["out"].clone(), d["attn"].clone(), d["st"][:2 * C].clone()
            calls["batch_stats"](d); calls["gap_t_bn"](d); calls["mix_logits_bn"](d)
            torch.cuda.synchronize()
            for nm, x, y in (("stats", m0, d["st"][:2 * C]), ("gap", g0, d["gap"]), ("out", o0, d["out"]), ("attn", a0, d["attn"])):
                ne = (x != y).sum().item()
                print(f"    {nm}: {ne} of {x.numel()} differ, max |d| {(x.float() - y.float()).abs().max().item():.3e}")


main()

"""cot_agg_forward_rowstats + cot_bn_rowstats_finalize against cot_agg_forward + cot_bn_batch_stats per CoTNet-50 stage shape, B = 80, bf16,
through the C ABI (HIP events on the launch stream, buffer sets rotated beyond the Infinity Cache)."""
import ctypes
import sys

import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from cotnet_amd import _lib  # noqa: E402

L = _lib.lib()
dev = torch.device("cuda")
P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None  # noqa: E731
BF = _lib.COT_BF16


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


for (N, C, H, W) in [(80, 64, 56, 56), (80, 128, 28, 28), (80, 256, 14, 14), (80, 512, 7, 7)]:
    wC, HW = C // 8, H * W
    geom = _lib.AggGeom(N, C, H, W, 1, wC, 3, 3, 1, 1, 1, 1, 1, 1)
    nws = int(L.cot_bn_act_workspace(N, C))
    nset = max(2, int(600e6 // (N * C * HW * 2 * 3.2)))
    sets = []
    for _ in range(nset):
        d = {"x": torch.randn(N, C, H, W, device=dev).bfloat16(), "w": (0.3 * torch.randn(N, 1, wC, 9, H, W, device=dev)).bfloat16(),
             "o": torch.empty(N, C, H, W, device=dev, dtype=torch.bfloat16), "rows": torch.empty(N * C * H * 2, device=dev),
             "st": torch.empty(2 * C + nws, device=dev), "rm": torch.zeros(C, device=dev), "rv": torch.ones(C, device=dev),
             "nbt": torch.zeros((), dtype=torch.int64, device=dev)}
        sets.append(d)
    calls = {
        "agg_forward": lambda d: L.cot_agg_forward(P(d["x"]), P(d["w"]), P(d["o"]), ctypes.byref(geom), BF, 0, None),
        "batch_stats": lambda d: L.cot_bn_batch_stats(P(d["o"]), P(d["st"]), P(d["st"][C:]), P(d["rm"]), P(d["rv"]), P(d["nbt"]), P(d["st"][2 * C:]), N, C, HW, 1e-5, 0.1, BF, None),
        "agg_forward_rowstats": lambda d: L.cot_agg_forward_rowstats(P(d["x"]), P(d["w"]), P(d["o"]), P(d["rows"]), None, None, None, None, 0, ctypes.byref(geom), BF, None),
        "rowstats_finalize": lambda d: L.cot_bn_rowstats_finalize(P(d["rows"]), P(d["st"]), P(d["st"][C:]), P(d["rm"]), P(d["rv"]), P(d["nbt"]), N, C, H, W, 1e-5, 0.1, None),
    }
    t = {}
    for nm, fn in calls.items():
        def chk(d, fn=fn, nm=nm):
            if fn(d):
                raise RuntimeError(nm + ": " + L.cot_last_error().decode())
        t[nm] = timeit(chk, sets)
    print(f"N{N} C{C} {H}x{W}: " + "  ".join(f"{k} {v:.1f}" for k, v in t.items()) +
          f"   | separate {t['agg_forward'] + t['batch_stats']:.1f} us -> epilogue {t['agg_forward_rowstats'] + t['rowstats_finalize']:.1f}")

"""What bounds the 1x1 kernel on the channel-major deep layers (one image of N*HW pixels): forward and data gradient per
layer shape of CoTNet-50's stages 3 / 4 at B = 80 with parts of the K loop switched off (cot_set_tuning key 24, C1LdsArgs::ablate; results
are wrong under ablation, times are what is measured): 1 no copies after the prologue, 2 no fragment reads, 4 one MFMA per step, 8 no
barriers, 16 no vmcnt waits, 31 all of them (= launch + prologue + epilogue), 32 no epilogue (nothing stored), 63 = 31 + 32 (launch + prologue
alone), 64 the accumulators stored directly (8 bytes per lane: correct results, no LDS round trip)."""
import ctypes
import sys

import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from cotnet_amd import _lib  # noqa: E402

L = _lib.lib()
dev = torch.device("cuda")
P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None  # noqa: E731
BF = _lib.COT_BF16
SHAPES = [("s3 conv1 1024->256", 1024, 256, 15680), ("s3 embed0 512->128", 512, 128, 15680), ("s3 conv1x1 256->256", 256, 256, 15680),
          ("s3 conv3 256->1024", 256, 1024, 15680), ("s4 conv1 2048->512", 2048, 512, 3920), ("s4 conv3 512->2048", 512, 2048, 3920),
          ("s2 conv3 128->512 (NCHW 28x28)", 128, 512, -784), ("s1 conv3 64->256 (NCHW 56x56)", 64, 256, -3136)]
ABL = [0, 1, 31, 32, 63, 64]
if len(sys.argv) > 1:
    ABL = [int(v) for v in sys.argv[1].split(',')]
for kv in (sys.argv[2].split(',') if len(sys.argv) > 2 else []):  # further tuning keys: KEY=VALUE,...
    assert L.cot_set_tuning(int(kv.split('=')[0]), int(kv.split('=')[1])) == 0


def timeit(fn, n=20):
    for i in range(3):
        fn(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n):
        fn(i)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


print(f"{'layer':34s} " + " ".join(f"{'abl ' + str(a):>9s}" for a in ABL) + "   (us; forward / data gradient)")
for name, Ci, Co, HW in SHAPES:
    N = 1
    if HW < 0:
        N, HW = 80, -HW
    nset = max(2, min(6, int(300e6 // ((Ci + Co) * N * HW * 2)) + 1))
    sets = [(torch.randn(N, Ci, HW, device=dev).bfloat16(), torch.randn(N, Co, HW, device=dev).bfloat16(),
             torch.empty(N, Co, HW, device=dev).bfloat16(), torch.empty(N, Ci, HW, device=dev).bfloat16()) for _ in range(nset)]
    w = (torch.randn(Co, Ci, device=dev) / Ci ** 0.5).bfloat16()
    ws = torch.empty(int(L.cot_conv1x1_workspace(N, Ci, Co, HW, 0)), dtype=torch.uint8, device=dev)
    res = []
    for a in ABL:
        assert L.cot_set_tuning(24, a) == 0

        def fwd(i):
            x, gy, y, gx = sets[i % nset]
            assert L.cot_conv1x1_forward(P(x), None, Ci, P(w), None, P(y), N, Ci, Co, HW, BF, None) == 0, L.cot_last_error()

        def dgr(i):
            x, gy, y, gx = sets[i % nset]
            assert L.cot_conv1x1_backward_data(P(gy), P(w), P(gx), None, Ci, 0, P(ws), N, Ci, Co, HW, BF, None) == 0, L.cot_last_error()
        res.append((timeit(fwd), timeit(dgr)))
    L.cot_set_tuning(24, 0)
    print(f"{name:34s} " + " ".join(f"{f:4.1f}/{d:4.1f}" for f, d in res))

"""Run the MFMA convolution / GroupNorm kernels (host-emulated) on tensors whose LAST byte sits right in front of an
inaccessible page: any read or write past the end of a tensor is a segmentation fault, not a silent success.
Executed in a child process by tests/test_kernels_emulated.py::test_no_kernel_touches_memory_past_its_tensors."""
import ctypes
import mmap
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from cotnet_amd import _lib  # noqa: E402
from tests.emul import build_emul  # noqa: E402

E = ctypes.CDLL(build_emul.build())
for _n, (_r, _a) in _lib.SYMBOLS.items():
    getattr(E, _n).restype, getattr(E, _n).argtypes = _r, _a
libc = ctypes.CDLL(None, use_errno=True)
PAGE = mmap.PAGESIZE
_keep = []


def guarded(t):
    """copy of t whose storage ends exactly at a PROT_NONE page (start stays 16-byte aligned: sizes here are multiples of 16
    or the tensor is padded at the FRONT)"""
    nbytes = t.numel() * t.element_size()
    span = (nbytes + PAGE - 1) // PAGE * PAGE + PAGE
    m = mmap.mmap(-1, span)
    addr = ctypes.addressof(ctypes.c_char.from_buffer(m))
    assert libc.mprotect(ctypes.c_void_p(addr + span - PAGE), PAGE, 0) == 0
    off = span - PAGE - nbytes
    assert off % 2 == 0
    g = torch.frombuffer(m, dtype=t.dtype, count=t.numel(), offset=off).view(t.shape)
    g.copy_(t)
    _keep.append(m)
    return g


def P(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


BF = _lib.dtype_code(torch.bfloat16)


def conv1x1(N, Ci, Co, H, W, c1):
    HW = H * W
    x = torch.randn(N, Ci, H, W).bfloat16()
    xs = [guarded(x[:, :c1].contiguous()), guarded(x[:, c1:].contiguous())] if c1 else [guarded(x), None]
    w, b = guarded(torch.randn(Co, Ci).bfloat16()), guarded(torch.randn(Co).bfloat16())
    y, gy = guarded(torch.empty(N, Co, H, W).bfloat16()), guarded(torch.randn(N, Co, H, W).bfloat16())
    cc1 = c1 or Ci
    assert E.cot_conv1x1_forward(P(xs[0]), P(xs[1]), cc1, P(w), P(b), P(y), N, Ci, Co, HW, BF, None) == 0
    ws = torch.empty(E.cot_conv1x1_workspace(N, Ci, Co, HW, 1), dtype=torch.uint8)
    gx = [guarded(torch.zeros_like(t)) if t is not None else None for t in xs]
    assert E.cot_conv1x1_backward_data(P(gy), P(w), P(gx[0]), P(gx[1]), cc1, 3, P(ws), N, Ci, Co, HW, BF, None) == 0
    gw, gb = guarded(torch.empty_like(w)), guarded(torch.empty_like(b))
    assert E.cot_conv1x1_backward_weight(P(gy), P(xs[0]), P(xs[1]), cc1, P(gw), P(gb), P(ws), N, Ci, Co, HW, BF, None) == 0


def conv3x3(N, C, G, H, W):
    x, gy = guarded(torch.randn(N, C, H, W).bfloat16()), guarded(torch.randn(N, C, H, W).bfloat16())
    w = guarded(torch.randn(C, C // G, 3, 3).bfloat16())
    masks = torch.empty(E.cot_conv3x3g_masks_bytes(H, W), dtype=torch.uint8)
    assert E.cot_conv3x3g_masks(P(masks), H, W, None) == 0
    ws = torch.empty(E.cot_conv3x3g_workspace(N, C, C, G, H, W), dtype=torch.uint8)
    y, gx, gw = guarded(torch.empty_like(x)), guarded(torch.zeros_like(x)), guarded(torch.empty_like(w))
    assert E.cot_conv3x3g_forward(P(x), P(w), P(y), P(masks), P(ws), N, C, C, G, H, W, BF, None) == 0
    assert E.cot_conv3x3g_backward_data(P(gy), P(w), P(gx), 1, P(masks), P(ws), N, C, C, G, H, W, BF, None) == 0
    assert E.cot_conv3x3g_backward_weight(P(gy), P(x), P(gw), P(masks), P(ws), N, C, C, G, H, W, BF, None) == 0


def conv_general(N, Ci, Co, G, H, W, ksize, dtype):
    """csrc/conv_gen.hip through cot_conv1x1g_* (ksize 1) / cot_conv3x3g_* (ksize 3, fp32 or channel counts off the grid)"""
    dt = _lib.dtype_code(dtype)
    x, gy = guarded(torch.randn(N, Ci, H, W).to(dtype)), guarded(torch.randn(N, Co, H, W).to(dtype))
    w, b = guarded(torch.randn(Co, Ci // G, ksize, ksize).to(dtype)), guarded(torch.randn(Co).to(dtype))
    y, gx, gw, gb = (guarded(torch.zeros_like(t)) for t in (gy, x, w, b))
    if ksize == 1:
        ws = guarded(torch.empty(E.cot_convg_workspace(N, Ci, Co, G, H * W, 1, 1), dtype=torch.uint8))
        assert E.cot_conv1x1g_forward(P(x), P(w), P(b), P(y), N, Ci, Co, G, H * W, dt, None) == 0
        assert E.cot_conv1x1g_backward_data(P(gy), P(w), P(gx), 1, N, Ci, Co, G, H * W, dt, None) == 0
        assert E.cot_conv1x1g_backward_weight(P(gy), P(x), P(gw), P(gb), P(ws), N, Ci, Co, G, H * W, dt, None) == 0
    else:
        masks = torch.empty(E.cot_conv3x3g_masks_bytes(H, W), dtype=torch.uint8)
        assert E.cot_conv3x3g_masks(P(masks), H, W, None) == 0
        ws = guarded(torch.empty(max(E.cot_conv3x3g_workspace(N, Ci, Co, G, H, W), E.cot_convg_workspace(N, Ci, Co, G, H, W, 3)),
                                 dtype=torch.uint8))
        assert E.cot_conv3x3g_forward(P(x), P(w), P(y), P(masks), P(ws), N, Ci, Co, G, H, W, dt, None) == 0
        assert E.cot_conv3x3g_backward_data(P(gy), P(w), P(gx), 1, P(masks), P(ws), N, Ci, Co, G, H, W, dt, None) == 0
        assert E.cot_conv3x3g_backward_weight(P(gy), P(x), P(gw), P(masks), P(ws), N, Ci, Co, G, H, W, dt, None) == 0


def gn9f(N, G, H, W):
    C = 9 * G
    x, dy = guarded(torch.randn(N, C, H, W)), guarded(torch.randn(N, C, H, W))
    ga, be = guarded(torch.randn(C)), guarded(torch.randn(C))
    y, dx = guarded(torch.empty_like(x)), guarded(torch.empty_like(x))
    mean, rstd = guarded(torch.empty(N * G)), guarded(torch.empty(N * G))
    dg, db, ws = guarded(torch.empty_like(ga)), guarded(torch.empty_like(ga)), guarded(torch.empty(2 * N * C))
    assert E.cot_group_norm9_forward(P(x), P(ga), P(be), P(y), P(mean), P(rstd), N, C, H * W, 1e-5, 0, None) == 0
    assert E.cot_group_norm9_backward(P(dy), P(x), P(mean), P(rstd), P(ga), P(dx), P(dg), P(db), P(ws), N, C, H * W, 0,
                                      None) == 0


def gn9(N, G, H, W):
    C, HW = 9 * G, H * W
    x, dy = guarded(torch.randn(N, C, H, W).bfloat16()), guarded(torch.randn(N, C, H, W).bfloat16())
    ga, be = guarded(torch.randn(C).bfloat16()), guarded(torch.randn(C).bfloat16())
    y, dx = guarded(torch.empty_like(x)), guarded(torch.empty_like(x))
    mean, rstd = torch.empty(N * G), torch.empty(N * G)
    assert E.cot_group_norm9_forward(P(x), P(ga), P(be), P(y), P(mean), P(rstd), N, C, HW, 1e-5, BF, None) == 0
    dg, db, ws = guarded(torch.empty_like(ga)), guarded(torch.empty_like(ga)), torch.empty(2 * N * C)
    assert E.cot_group_norm9_backward(P(dy), P(x), P(mean), P(rstd), P(ga), P(dx), P(dg), P(db), P(ws), N, C, HW, BF,
                                      None) == 0


def aggregation(N, C, H, W, dtype):
    import ctypes as ct
    wC = C // 8
    x, w = guarded(torch.randn(N, C, H, W).to(dtype)), guarded(torch.randn(N, 1, wC, 9, H, W).to(dtype))
    out, g = guarded(torch.empty(N, C, H, W).to(dtype)), guarded(torch.randn(N, C, H, W).to(dtype))
    gx, gw = guarded(torch.empty_like(x)), guarded(torch.empty_like(w))
    geom = _lib.AggGeom(N, C, H, W, 1, wC, 3, 3, 1, 1, 1, 1, 1, 1)
    dt = _lib.dtype_code(dtype)
    assert E.cot_agg_forward(P(x), P(w), P(out), ct.byref(geom), dt, 0, None) == 0
    assert E.cot_agg_backward(P(g), P(x), P(w), P(gx), P(gw), ct.byref(geom), dt, 0, None) == 0


def bn_act(N, C, H, W, dtype, fold):
    assert E.cot_set_tuning(12, fold) == 0
    HW = H * W
    x, res, dy = (guarded(torch.randn(N, C, H, W).to(dtype)) for _ in range(3))
    y, dx, dres = (guarded(torch.empty(N, C, H, W).to(dtype)) for _ in range(3))
    ga, be, mean, rstd, rm, rv, dg, db = (guarded(torch.ones(C)) for _ in range(8))
    nbt = torch.zeros((), dtype=torch.int64)
    ws = torch.empty(E.cot_bn_act_workspace(N, C))
    dt = _lib.dtype_code(dtype)
    assert E.cot_bn_act_forward(P(x), P(res), P(y), P(ga), P(be), P(mean), P(rstd), P(rm), P(rv), P(nbt), P(ws), N, C, HW,
                                1e-5, 0.1, 1, dt, None) == 0
    assert E.cot_bn_act_backward(P(dy), P(x), P(y), P(dx), P(dres), P(ga), P(be), P(mean), P(rstd), P(dg), P(db), P(ws), N,
                                 C, HW, 1, dt, None) == 0
    assert E.cot_set_tuning(12, 0) == 0


def radix(N, C, H, W, dtype):
    HW = H * W
    dt = _lib.dtype_code(dtype)
    y, k, g = (guarded(torch.randn(N, C, H, W).to(dtype)) for _ in range(3))
    out, gy, gk = (guarded(torch.empty(N, C, H, W).to(dtype)) for _ in range(3))
    gapT, ggapT = guarded(torch.empty(C, N).to(dtype)), guarded(torch.randn(C, N).to(dtype))
    logT, glogT = guarded(torch.randn(2 * C, N).to(dtype)), guarded(torch.empty(2 * C, N).to(dtype))
    attn = guarded(torch.empty(N, C, 2).to(dtype))
    assert E.cot_radix_gap_t(P(y), P(k), P(gapT), N, C, HW, dt, None) == 0
    assert E.cot_radix_mix_logits(P(y), P(k), P(logT), P(out), P(attn), N, C, HW, dt, None) == 0
    assert E.cot_radix_mix_backward_reduce(P(g), P(y), P(k), P(attn), P(glogT), N, C, HW, dt, None) == 0
    assert E.cot_radix_mix_backward_apply(P(g), P(attn), P(ggapT), P(gy), P(gk), N, C, HW, dt, None) == 0
    gap, ga = guarded(torch.empty(N, C).to(dtype)), guarded(torch.empty(N, C, 2).to(dtype))
    lg, glg = guarded(torch.randn(N, C).to(dtype)), guarded(torch.empty(N, C).to(dtype))
    assert E.cot_se_gap(P(y), P(gap), N * C, HW, dt, None) == 0
    assert E.cot_se_gate(P(y), P(lg), P(out), N * C, HW, dt, None) == 0
    assert E.cot_se_gate_backward(P(g), P(y), P(lg), P(gy), P(glg), N * C, HW, dt, None) == 0
    assert E.cot_radix_gap(P(y), P(k), P(gap), N * C, HW, dt, None) == 0
    assert E.cot_radix_mix(P(y), P(k), P(attn), P(out), N * C, HW, dt, None) == 0
    assert E.cot_radix_mix_backward(P(g), P(y), P(k), P(attn), P(gy), P(gk), P(ga), N * C, HW, dt, None) == 0


def pooling(N, C, H, W, dtype):
    dt = _lib.dtype_code(dtype)
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    x, gx = guarded(torch.randn(N, C, H, W).to(dtype)), guarded(torch.empty(N, C, H, W).to(dtype))
    y, gy = guarded(torch.empty(N, C, Ho, Wo).to(dtype)), guarded(torch.randn(N, C, Ho, Wo).to(dtype))
    assert E.cot_maxpool3x3s2_forward(P(x), P(y), N * C, H, W, dt, None) == 0
    assert E.cot_maxpool3x3s2_backward(P(gy), P(x), P(gx), N * C, H, W, dt, None) == 0
    taps = guarded(torch.empty(y.shape, dtype=torch.uint8))
    assert E.cot_maxpool3x3s2_forward_taps(P(x), P(y), P(taps), N * C, H, W, dt, None) == 0
    assert E.cot_maxpool3x3s2_backward_taps(P(gy), P(taps), P(gx), N * C, H, W, dt, None) == 0
    if H >= 2 and W >= 2:
        assert E.cot_blurpool3x3s2_forward(P(x), P(y), N * C, H, W, dt, None) == 0
        assert E.cot_blurpool3x3s2_backward(P(gy), P(gx), N * C, H, W, dt, None) == 0
    assert E.cot_avgpool3x3s2_forward(P(x), P(y), N * C, H, W, dt, None) == 0
    assert E.cot_avgpool3x3s2_backward(P(gy), P(gx), N * C, H, W, dt, None) == 0


def stem(N, H, W):
    x, w = guarded(torch.randn(N, 3, H, W).bfloat16()), guarded(torch.randn(64, 3, 7, 7).bfloat16())
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    y, gy = (guarded(torch.randn(N, 64, Ho, Wo).bfloat16()) for _ in range(2))
    gw = guarded(torch.empty_like(w))
    ws = torch.empty(E.cot_stem7x7s2_workspace(N, H, W), dtype=torch.uint8)
    assert E.cot_stem7x7s2_forward(P(x), P(w), P(y), N, H, W, BF, None) == 0
    assert E.cot_stem7x7s2_backward_weight(P(gy), P(x), P(gw), P(ws), N, H, W, BF, None) == 0


if __name__ == "__main__":
    torch.manual_seed(0)
    E.cot_set_tuning(17, 8)  # also the general LDS weight-gradient kernel (2-byte-aligned DMA sources, tensor-end path)
    for shape in [(8, 16, 80, 7, 7, 0), (3, 64, 72, 14, 14, 32), (2, 40, 24, 6, 6, 0)]:
        conv1x1(*shape)
    E.cot_set_tuning(17, 0)
    # tensors whose byte size is a multiple of 16 keep the 16-byte base alignment the ABI asks for
    for shape in [(2, 64, 32, 8, 16, 0), (1, 40, 72, 12, 12, 0), (2, 32, 24, 14, 14, 0), (8, 16, 80, 7, 7, 0),
                  (2, 48, 40, 8, 8, 16), (8, 24, 16, 7, 7, 8), (8, 8, 8, 1, 1, 0),
                  # K % 32 == 0: the LDS-pipelined kernels (csrc/conv_lds.hip): partial pixel tile, flat images, partial
                  # image group, two slabs, channel counts that are not multiples of the tile
                  (1, 64, 40, 20, 20, 0), (3, 64, 72, 14, 14, 32), (8, 32, 24, 7, 7, 0), (1, 96, 136, 1, 80, 0), (1, 40, 24, 1, 24, 0),
                  (1, 16, 8, 3, 8, 0),
                  (2, 32, 32, 16, 24, 0)]:
        conv1x1(*shape)
    for shape in [(2, 64, 4, 8, 16), (2, 32, 4, 14, 14), (8, 64, 8, 7, 7), (1, 256, 4, 5, 4), (8, 16, 2, 1, 3),
                  # LDS kernels: BIG with row tiles whose halo runs past both ends of the tensor, FLAT with a partial image group
                  (1, 128, 4, 30, 20), (1, 64, 4, 24, 24), (3, 256, 4, 7, 7),
                  # group widths padded to whole 32-channel chunks: the LAST group's padding lies behind the tensor (clamped copies)
                  (1, 96, 2, 24, 16), (2, 192, 8, 8, 8), (3, 384, 8, 7, 7)]:
        conv3x3(*shape)
    # grouped 1x1 convolutions group by group on the tuned kernels (image strides of the full tensors), weight gradient
    # with 12-wide groups merged in pairs
    for shape in [(2, 128, 96, 2, 8, 8, 1), (1, 64, 48, 2, 20, 18, 1), (2, 96, 96, 8, 6, 10, 3)]:
        conv_general(*shape, torch.bfloat16)
    for shape in [(2, 2, 8, 8), (2, 1, 14, 14), (8, 2, 7, 7), (1, 1, 56, 56), (8, 2, 3, 5)]:
        gn9(*shape)
    for dtype in (torch.float32, torch.bfloat16):
        # (element counts are multiples of 8 so that the guarded tensors keep the 16-byte base alignment the ABI asks for)
        for shape in [(2, 32, 24, 1, 4, 4, 1), (2, 48, 108, 2, 4, 6, 1), (1, 16, 8, 2, 1, 8, 1), (2, 16, 16, 4, 4, 4, 3),
                      (1, 96, 96, 8, 2, 4, 3), (2, 72, 72, 1, 2, 2, 3)]:
            conv_general(*shape, dtype)
    for shape in [(2, 2, 8, 8), (8, 2, 2, 2), (1, 1, 56, 56), (4, 1, 2, 6)]:
        gn9f(*shape)
    for shape in [(2, 32, 32), (1, 16, 64)]:
        stem(*shape)
    for dtype in (torch.bfloat16, torch.float32):
        for shape in [(2, 16, 6, 56), (2, 16, 5, 28), (2, 32, 14, 14), (8, 64, 7, 7), (8, 8, 3, 10)]:
            aggregation(*shape, dtype)
        for shape in [(8, 8, 7, 7), (4, 16, 14, 14), (2, 8, 8, 8), (8, 8, 1, 1)]:
            for fold in (0, 1):
                for chan in (0, 1):  # streaming kernels, then the channel-resident ones
                    E.cot_set_tuning(21, chan)
                    bn_act(*shape, dtype, fold)
            radix(*shape, dtype)
        for shape in [(2, 4, 8, 8), (1, 8, 7, 7), (2, 4, 5, 9), (8, 2, 1, 1)]:
            pooling(*shape, dtype)
    print("GUARD_OK")

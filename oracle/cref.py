"""ctypes front-end of oracle/liboracle.so -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module.  It wraps the C restatement in agg_oracle.c (which cites the
reference lines it follows) for CPU torch tensors, fp32 / fp64, NCHW.
"""
import ctypes
import os
import subprocess

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class _Geom(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int) for n in (
        "N", "C", "H", "W", "heads", "wC", "kh", "kw", "sh", "sw", "ph", "pw", "dh", "dw", "Ho", "Wo")]


class _MixGeom(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int) for n in (
        "N", "C", "H", "W", "heads", "wC", "sh", "sw", "dh", "dw", "p1h", "p1w", "p2h", "p2w", "Ho", "Wo")]


def build(force=False):
    so = os.path.join(_HERE, "liboracle.so")
    src = os.path.join(_HERE, "agg_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "liboracle.so"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = ctypes.CDLL(build())
        _LIB.agg_oracle_out_size.restype = ctypes.c_int
    return _LIB


def _pair(v):
    return (int(v), int(v)) if isinstance(v, int) else (int(v[0]), int(v[1]))


def out_size(n, k, s, p, d):
    return lib().agg_oracle_out_size(int(n), int(k), int(s), int(p), int(d))


def _suffix(t):
    if t.dtype == torch.float32:
        return "f32"
    if t.dtype == torch.float64:
        return "f64"
    raise TypeError(f"oracle supports float32/float64 only, got {t.dtype}")


def _ptr(t):
    assert t.device.type == "cpu" and t.is_contiguous()
    return ctypes.c_void_p(t.data_ptr())


def geometry(x_shape, w_shape, kernel_size, stride, padding, dilation):
    k, s, p, d = _pair(kernel_size), _pair(stride), _pair(padding), _pair(dilation)
    N, C, H, W = x_shape
    _, heads, wC, taps, wH, wW = w_shape
    Ho, Wo = out_size(H, k[0], s[0], p[0], d[0]), out_size(W, k[1], s[1], p[1], d[1])
    assert taps == k[0] * k[1] and Ho * Wo == wH * wW and C % wC == 0
    return _Geom(N, C, H, W, heads, wC, k[0], k[1], s[0], s[1], p[0], p[1], d[0], d[1], Ho, Wo)


def forward(x, w, kernel_size=3, stride=1, padding=0, dilation=1):
    x, w = x.contiguous(), w.contiguous()
    g = geometry(x.shape, w.shape, kernel_size, stride, padding, dilation)
    out = torch.empty(g.N, g.heads * g.C, g.Ho, g.Wo, dtype=x.dtype)
    getattr(lib(), "agg_oracle_forward_" + _suffix(x))(_ptr(x), _ptr(w), _ptr(out), ctypes.byref(g))
    return out


def backward_input(gout, w, x_shape, kernel_size=3, stride=1, padding=0, dilation=1):
    gout, w = gout.contiguous(), w.contiguous()
    g = geometry(x_shape, w.shape, kernel_size, stride, padding, dilation)
    gx = torch.empty(*x_shape, dtype=gout.dtype)
    getattr(lib(), "agg_oracle_backward_input_" + _suffix(gout))(_ptr(gout), _ptr(w), _ptr(gx), ctypes.byref(g))
    return gx


def backward_weight(gout, x, w_shape, kernel_size=3, stride=1, padding=0, dilation=1):
    gout, x = gout.contiguous(), x.contiguous()
    g = geometry(x.shape, w_shape, kernel_size, stride, padding, dilation)
    gw = torch.empty(*w_shape, dtype=gout.dtype)
    getattr(lib(), "agg_oracle_backward_weight_" + _suffix(gout))(_ptr(gout), _ptr(x), _ptr(gw), ctypes.byref(g))
    return gw


# ---- mix (3x3 + 5x5) -------------------------------------------------------
def mix_geometry(x_shape, w1_shape, w2_shape, stride, padding1, padding2, dilation):
    s, p1, p2, d = _pair(stride), _pair(padding1), _pair(padding2), _pair(dilation)
    N, C, H, W = x_shape
    _, heads, wC, taps1, wH, wW = w1_shape
    assert taps1 == 9 and w2_shape[3] == 25 and tuple(w2_shape[:3]) == tuple(w1_shape[:3])
    # output size is computed from the 3x3 set only (mix.py:216-217)
    Ho, Wo = out_size(H, 3, s[0], p1[0], d[0]), out_size(W, 3, s[1], p1[1], d[1])
    assert Ho * Wo == wH * wW
    return _MixGeom(N, C, H, W, heads, wC, s[0], s[1], d[0], d[1], p1[0], p1[1], p2[0], p2[1], Ho, Wo)


def mix_forward(x, w1, w2, stride=1, padding1=0, padding2=0, dilation=1):
    x, w1, w2 = x.contiguous(), w1.contiguous(), w2.contiguous()
    g = mix_geometry(x.shape, w1.shape, w2.shape, stride, padding1, padding2, dilation)
    out = torch.empty(g.N, 2 * g.heads * g.C, g.Ho, g.Wo, dtype=x.dtype)
    getattr(lib(), "aggmix_oracle_forward_" + _suffix(x))(_ptr(x), _ptr(w1), _ptr(w2), _ptr(out), ctypes.byref(g))
    return out


def mix_backward_input(gout, w1, w2, x_shape, stride=1, padding1=0, padding2=0, dilation=1, all_heads=False):
    gout, w1, w2 = gout.contiguous(), w1.contiguous(), w2.contiguous()
    g = mix_geometry(x_shape, w1.shape, w2.shape, stride, padding1, padding2, dilation)
    gx = torch.empty(*x_shape, dtype=gout.dtype)
    getattr(lib(), "aggmix_oracle_backward_input_" + _suffix(gout))(
        _ptr(gout), _ptr(w1), _ptr(w2), _ptr(gx), ctypes.byref(g), ctypes.c_int(1 if all_heads else 0))
    return gx


def mix_backward_weight(gout, x, w1_shape, w2_shape, stride=1, padding1=0, padding2=0, dilation=1):
    gout, x = gout.contiguous(), x.contiguous()
    g = mix_geometry(x.shape, w1_shape, w2_shape, stride, padding1, padding2, dilation)
    gw1 = torch.empty(*w1_shape, dtype=gout.dtype)
    gw2 = torch.empty(*w2_shape, dtype=gout.dtype)
    getattr(lib(), "aggmix_oracle_backward_weight_" + _suffix(gout))(
        _ptr(gout), _ptr(x), _ptr(gw1), _ptr(gw2), ctypes.byref(g))
    return gw1, gw2

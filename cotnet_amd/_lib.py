"""ctypes binding of libcotnet_hip.so (the C ABI in include/cotnet_amd.h).

This is the only place the shared library is opened.  There is NO fallback: if the library is missing
or a call fails, a RuntimeError is raised (the product path must fail loudly, never silently compute
on the CPU).  The reference's equivalent is cupy_layers/utils.py:14-18 (`load_kernel`).
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# COT_LIB_PATH: developer A/B of two builds of the library on one box (scripts/gpu_session.sh); the product loads the in-tree one
LIB_PATH = os.environ.get("COT_LIB_PATH") or os.path.join(_HERE, "lib", "libcotnet_hip.so")

COT_F32, COT_F64, COT_BF16, COT_F16 = 0, 1, 2, 3
COT_NCHW, COT_NHWC = 0, 1
COT_ERR_UNSUPPORTED = -2  # (cot_status, include/cotnet_amd.h)

_lib = None
_SHAPE_CACHES = []  # dicts of shape -> workspace size held by the Python wrappers (see register_cache)


def register_cache(d):
    """Workspace sizes are pure functions of (shape, tuning state).  The wrappers cache them per shape; every cache is
    registered here and emptied whenever a tuning key changes (several keys -- 11, 15, 17, 19, 20 -- change what
    cot_conv1x1_workspace / cot_conv3x3g_workspace return: a stale, smaller size would let a weight-gradient kernel write
    its partial sums past the end of the workspace; the ABI has no size argument to catch that)."""
    _SHAPE_CACHES.append(d)
    return d


class AggGeom(ctypes.Structure):
    """mirror of `cot_agg_geom`"""
    _fields_ = [(n, ctypes.c_int32) for n in (
        "N", "C", "H", "W", "heads", "wC", "kh", "kw", "sh", "sw", "ph", "pw", "dh", "dw")]


class ProfileRec(ctypes.Structure):
    """mirror of `cot_profile_rec`"""
    _fields_ = [("kernel", ctypes.c_char * 48), ("kind", ctypes.c_int32), ("flags", ctypes.c_int32),
                ("dtype", ctypes.c_int32), ("layout", ctypes.c_int32), ("geom", AggGeom), ("ms", ctypes.c_float)]


# every symbol include/cotnet_amd.h declares: (restype, argtypes)
_P, _I = ctypes.c_void_p, ctypes.c_int
_G = ctypes.POINTER(AggGeom)
SYMBOLS = {
    "cot_abi_version": (_I, []),
    "cot_last_error": (ctypes.c_char_p, []),
    "cot_last_kernel": (ctypes.c_char_p, []),
    "cot_status_string": (ctypes.c_char_p, [_I]),
    "cot_agg_out_size": (_I, [_I] * 5),
    "cot_agg_forward": (_I, [_P, _P, _P, _G, _I, _I, _P]),
    "cot_agg_backward_input": (_I, [_P, _P, _P, _G, _I, _I, _P]),
    "cot_agg_backward_weight": (_I, [_P, _P, _P, _G, _I, _I, _P]),
    "cot_agg_backward": (_I, [_P, _P, _P, _P, _P, _G, _I, _I, _P]),
    "cot_agg_softmax_forward": (_I, [_P, _P, _P, _P, _G, _I, _P]),
    "cot_agg_softmax_backward": (_I, [_P, _P, _P, _P, _P, _G, _I, _P]),
    "cot_aggmix_forward": (_I, [_P, _P, _P, _P, _G, _I, _I, _I, _P]),
    "cot_aggmix_backward_input": (_I, [_P, _P, _P, _P, _G, _I, _I, _I, _I, _P]),
    "cot_aggmix_backward_weight": (_I, [_P, _P, _P, _P, _G, _I, _I, _I, _P]),
    "cot_set_tuning": (_I, [_I, _I]),
    "cot_xchg_mode": (_I, []),
    "cot_launch_log": (_I, [ctypes.c_char_p, _I]),
    "cot_radix_gap": (_I, [_P, _P, _P, ctypes.c_int64, _I, _I, _P]),
    "cot_radix_mix": (_I, [_P, _P, _P, _P, ctypes.c_int64, _I, _I, _P]),
    "cot_radix_mix_backward": (_I, [_P] * 7 + [ctypes.c_int64, _I, _I, _P]),
    "cot_radix_gap_t": (_I, [_P, _P, _P, _I, _I, _I, _I, _P]),
    "cot_radix_mix_logits": (_I, [_P] * 5 + [_I, _I, _I, _I, _P]),
    "cot_radix_mix_backward_reduce": (_I, [_P] * 5 + [_I, _I, _I, _I, _P]),
    "cot_radix_mix_backward_apply": (_I, [_P] * 5 + [_I, _I, _I, _I, _P]),
    "cot_group_norm9_forward": (_I, [_P] * 6 + [_I, _I, _I, ctypes.c_float, _I, _P]),
    "cot_group_norm9_backward": (_I, [_P] * 9 + [_I, _I, _I, _I, _P]),
    "cot_gn9_stats_floats": (ctypes.c_int64, [_I, _I, _I]),
    "cot_gn9_fused_covers": (_I, [_I] * 5),
    "cot_conv1x1_forward_gn9": (_I, [_P, _P, _I, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "cot_gn9_stats_finalize": (_I, [_P, _P, _P, _I, _I, _I, ctypes.c_float, _P]),
    "cot_agg_gn9_forward": (_I, [_P] * 6 + [_I, _P, _G, _I, _P]),
    "cot_agg_gn9_backward": (_I, [_P] * 7 + [_I, _P, _P, _G, _I, _P]),
    "cot_subsample2_forward": (_I, [_P, _P, ctypes.c_int64, _I, _I, _I, _P]),
    "cot_subsample2_backward": (_I, [_P, _P, ctypes.c_int64, _I, _I, _I, _P]),
    "cot_avgpool2x2s2_forward": (_I, [_P, _P, ctypes.c_int64, _I, _I, _I, _P]),
    "cot_avgpool2x2s2_backward": (_I, [_P, _P, ctypes.c_int64, _I, _I, _I, _P]),
    "cot_avgpool3x3s2_forward": (_I, [_P, _P, ctypes.c_int64, _I, _I, _I, _P]),
    "cot_avgpool3x3s2_backward": (_I, [_P, _P, ctypes.c_int64, _I, _I, _I, _P]),
    "cot_maxpool3x3s2_forward": (_I, [_P, _P, ctypes.c_int64, _I, _I, _I, _P]),
    "cot_maxpool3x3s2_backward": (_I, [_P, _P, _P, ctypes.c_int64, _I, _I, _I, _P]),
    "cot_blurpool3x3s2_forward": (_I, [_P, _P, ctypes.c_int64, _I, _I, _I, _P]),
    "cot_blurpool3x3s2_backward": (_I, [_P, _P, ctypes.c_int64, _I, _I, _I, _P]),
    "cot_se_gap": (_I, [_P, _P, ctypes.c_int64, _I, _I, _P]),
    "cot_se_gate": (_I, [_P, _P, _P, ctypes.c_int64, _I, _I, _P]),
    "cot_se_gate_backward": (_I, [_P, _P, _P, _P, _P, ctypes.c_int64, _I, _I, _P]),
    "cot_maxpool3x3s2_forward_taps": (_I, [_P, _P, _P, ctypes.c_int64, _I, _I, _I, _P]),
    "cot_maxpool3x3s2_backward_taps": (_I, [_P, _P, _P, ctypes.c_int64, _I, _I, _I, _P]),
    "cot_stem7x7s2_workspace": (ctypes.c_int64, [_I, _I, _I]),
    "cot_stem7x7s2_forward": (_I, [_P, _P, _P, _I, _I, _I, _I, _P]),
    "cot_stem7x7s2_backward_weight": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "cot_stem3x3s2_workspace": (ctypes.c_int64, [_I, _I, _I, _I]),
    "cot_stem3x3s2_forward": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "cot_stem3x3s2_backward_weight": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "cot_sgd_step": (_I, [_P, _P, _P, _P, ctypes.c_int64, ctypes.c_float, ctypes.c_float, ctypes.c_float, ctypes.c_float,
                          _I, _I, _I, _P]),
    "cot_conv1x1_workspace": (ctypes.c_int64, [_I, _I, _I, _I, _I]),
    "cot_conv1x1_forward": (_I, [_P, _P, _I, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "cot_conv1x1_backward_data": (_I, [_P, _P, _P, _P, _I, _I, _P, _I, _I, _I, _I, _I, _P]),
    "cot_conv1x1_backward_weight": (_I, [_P, _P, _P, _I, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "cot_conv1x1_backward_data_relu_res_covers": (_I, [_I] * 5),
    "cot_conv1x1_backward_data_relu_res": (_I, [_P] * 5 + [_I] * 5 + [_P]),
    "cot_conv3x3g_masks_bytes": (ctypes.c_int64, [_I, _I]),
    "cot_conv3x3g_masks": (_I, [_P, _I, _I, _P]),
    "cot_conv3x3g_workspace": (ctypes.c_int64, [_I] * 6),
    "cot_conv3x3g_forward": (_I, [_P] * 5 + [_I] * 7 + [_P]),
    "cot_conv3x3g_backward_data": (_I, [_P] * 3 + [_I] + [_P] * 2 + [_I] * 7 + [_P]),
    "cot_conv3x3g_backward_weight": (_I, [_P] * 5 + [_I] * 7 + [_P]),
    "cot_conv3x3g_backward_weight_guarded": (_I, [_P] * 5 + [_I] * 8 + [_P]),
    "cot_conv3x3g_packed_bytes": (ctypes.c_int64, [_I] * 3),
    "cot_conv3x3g_pack": (_I, [_P, _P] + [_I] * 8 + [_P]),
    "cot_conv3x3g_forward_packed": (_I, [_P] * 3 + [_I] * 7 + [_P]),
    "cot_conv3x3g_backward_data_packed": (_I, [_P] * 3 + [_I] * 8 + [_P]),
    "cot_convg_workspace": (ctypes.c_int64, [_I] * 7),
    "cot_conv1x1g_forward": (_I, [_P] * 4 + [_I] * 6 + [_P]),
    "cot_conv1x1g_backward_data": (_I, [_P] * 3 + [_I] * 7 + [_P]),
    "cot_conv1x1g_backward_weight": (_I, [_P] * 5 + [_I] * 6 + [_P]),
    "cot_ema_step": (_I, [_P, _P, ctypes.c_int64, ctypes.c_float, _I, _P]),
    "cot_conv1x1_lds_covers": (_I, [_I, _I, _I, _I]),
    "cot_input_normalize": (_I, [_P, _P, _P, _P, ctypes.c_int64, _I, _I, _I, _P]),
    "cot_bn_act_workspace": (_I, [_I, _I]),
    "cot_bn_act_forward": (_I, [_P] * 11 + [_I, _I, _I, ctypes.c_float, ctypes.c_float, _I, _I, _P]),
    "cot_bn_act_backward": (_I, [_P] * 12 + [_I, _I, _I, _I, _I, _P]),
    "cot_bn_act_forward_ps": (_I, [_P] * 12 + [_I, _I, _I, ctypes.c_float, ctypes.c_float, _I, _I, _P]),
    "cot_bn_act_backward_ps": (_I, [_P] * 13 + [_I, _I, _I, _I, _I, _P]),
    "cot_bn_relu_mask_bytes": (ctypes.c_int64, [_I, _I, _I, _I]),
    "cot_bn_act_forward_mask": (_I, [_P] * 13 + [_I, _I, _I, ctypes.c_float, ctypes.c_float, _I, _I, _P]),
    "cot_bn_act_backward_mask": (_I, [_P] * 13 + [_I, _I, _I, _I, _I, _P]),
    "cot_conv1x1_stats_covers": (_I, [_I] * 4),
    "cot_conv1x1_forward_stats": (_I, [_P, _P, _I, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "cot_bn_tile_stats_finalize": (_I, [_P] * 6 + [_I, _I, _I, ctypes.c_float, ctypes.c_float, _P]),
    "cot_bn_act_apply_forward": (_I, [_P] * 8 + [_I, _I, _I, _I, _I, _P]),
    "cot_bn_act_lay_covers": (_I, [_I, _I, _I, _I]),
    "cot_bn_act_forward_lay": (_I, [_P] * 12 + [_I, _I, _I, ctypes.c_float, ctypes.c_float, _I, _I, _I, _P]),
    "cot_bn_act_backward_lay": (_I, [_P] * 13 + [_I, _I, _I, _I, _I, _I, _P]),
    "cot_radix_gap_t_lay": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "cot_radix_mix_logits_lay": (_I, [_P] * 5 + [_I, _I, _I, _I, _I, _P]),
    "cot_radix_mix_backward_reduce_lay": (_I, [_P] * 5 + [_I, _I, _I, _I, _I, _P]),
    "cot_radix_mix_backward_apply_lay": (_I, [_P] * 5 + [_I, _I, _I, _I, _I, _P]),
    "cot_agg_rowstats_floats": (ctypes.c_int64, [_I, _I, _I]),
    "cot_agg_forward_rowstats": (_I, [_P] * 8 + [_I, _G, _I, _P]),
    "cot_bn_rowstats_finalize": (_I, [_P] * 6 + [_I, _I, _I, _I, ctypes.c_float, ctypes.c_float, _P]),
    "cot_bn_batch_stats": (_I, [_P] * 7 + [_I, _I, _I, ctypes.c_float, ctypes.c_float, _I, _P]),
    "cot_bn_stats_sums": (_I, [_P, _P, _I, _I, _I, _I, _P]),
    "cot_radix_gap_t_bn": (_I, [_P] * 11 + [_I, _I, _I, ctypes.c_float, ctypes.c_float, _I, _I, _P]),
    "cot_radix_mix_logits_bn": (_I, [_P] * 9 + [_I, _I, _I, _I, _I, _P]),
    "cot_radix_mix_backward_reduce_bn": (_I, [_P] * 10 + [_I, _I, _I, _I, _I, _P]),
    "cot_radix_mix_backward_apply_bn": (_I, [_P] * 13 + [_I, _I, _I, _I, _I, _P]),
    "cot_group_norm9_forward_lay": (_I, [_P] * 6 + [_I, _I, _I, ctypes.c_float, _I, _I, _P]),
    "cot_group_norm9_backward_lay": (_I, [_P] * 9 + [_I, _I, _I, _I, _I, _P]),
    "cot_group_norm9_backward_params": (_I, [_P] * 3 + [_I, _I, _I, _P]),
    "cot_bn_act_inference": (_I, [_P] * 7 + [_I, _I, _I, ctypes.c_float, _I, _I, _P]),
    "cot_profile_begin": (_I, []),
    "cot_profile_end": (_I, [ctypes.POINTER(ProfileRec), _I]),
}


def build(verbose=False):
    """Compile the HIP sources for gfx950 (used by __graft_entry__.build and by developers)."""
    import subprocess
    cmd = ["make", "-C", os.path.join(_HERE, "csrc"), "-j8"]
    subprocess.check_call(cmd, stdout=None if verbose else subprocess.DEVNULL)
    return LIB_PATH


# Module fallbacks.  Every wrapper (conv1x1, conv3x3g, fused_bn, group_norm9, pool3x3, stem7x7, head_fused) serves a tensor that is off its
# kernels' grid with the torch module it wraps.  While the library's kernels were asked for (the wrapper's switch is on) and the tensor is
# on a GPU, each such call is counted here per site (bench.py prints the counters as `module_fallbacks`), and COT_STRICT_DISPATCH=1
# turns it into an error -- for deployments that must never leave the hand-written path silently.
STRICT_DISPATCH = os.environ.get("COT_STRICT_DISPATCH", "0") == "1"
FALLBACKS = {}


def fallback(site, x=None, detail=""):
    if x is not None and not getattr(x, "is_cuda", False):
        return
    FALLBACKS[site] = FALLBACKS.get(site, 0) + 1
    if STRICT_DISPATCH:
        shape = tuple(x.shape) if x is not None else ()
        raise RuntimeError(f"{site}: {shape} {getattr(x, 'dtype', '')} {detail} is off the library's kernels and COT_STRICT_DISPATCH=1 "
                           "(the torch module would have run)")


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950). cotnet_amd has no CPU or eager fallback.")
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            if os.environ.get("COT_LIB_PATH") and not hasattr(L, name):
                continue  # (developer A/B against an OLDER build of the library: entry points it lacks are simply not bound)
            fn = getattr(L, name)  # AttributeError here = header/library mismatch
            fn.restype, fn.argtypes = res, args
        if L.cot_abi_version() != 1:
            raise RuntimeError(f"libcotnet_hip.so ABI version {L.cot_abi_version()} != 1")
        raw_set_tuning = L.cot_set_tuning

        def cot_set_tuning(key, value):  # every caller (tests, bench.py --tune, COT_TUNING) goes through this wrapper
            for d in _SHAPE_CACHES:
                d.clear()
            return raw_set_tuning(key, value)
        L.cot_set_tuning = cot_set_tuning
        # developer knobs: COT_TUNING="12=1,9=0" -> cot_set_tuning(12, 1), cot_set_tuning(9, 0)  (include/cotnet_amd.h)
        for item in filter(None, os.environ.get("COT_TUNING", "").split(",")):
            k, v = item.split("=")
            if L.cot_set_tuning(int(k), int(v)) != 0:
                raise RuntimeError(f"COT_TUNING: {L.cot_last_error().decode()}")
        _lib = L
    return _lib


def check(status, what):
    if status != 0:
        L = lib()
        raise RuntimeError(f"{what} failed: {L.cot_status_string(status).decode()} -- "
                           f"{L.cot_last_error().decode()}")


def last_kernel():
    return lib().cot_last_kernel().decode()


_DTYPES = None


def dtype_code(torch_dtype):
    global _DTYPES
    if _DTYPES is None:
        import torch
        _DTYPES = {torch.float32: COT_F32, torch.float64: COT_F64, torch.bfloat16: COT_BF16, torch.float16: COT_F16}
    try:
        return _DTYPES[torch_dtype]
    except KeyError:
        raise TypeError(f"cotnet_amd: unsupported dtype {torch_dtype} (float32/float64/bfloat16/float16)")

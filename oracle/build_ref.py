"""oracle/build_ref.py -- run the REFERENCE's own kernel source on the CPU.  TEST INFRASTRUCTURE.

The reference keeps its device code as CUDA-C strings inside
cupy_layers/aggregation_zeropad.py (:20-110) and aggregation_zeropad_mix.py
(:20-207) and JIT-compiles them with CuPy/NVRTC after substituting every shape
as a literal (cupy_layers/utils.py:14-18).  Neither CuPy nor CUDA exist here, but
the kernel bodies are plain C.  This script

  1. imports the reference modules *where they lie* under $COT_REFERENCE
     (default /root/reference) with a stub `cupy` module, and reads the kernel
     source strings from them (nothing is copied into this repository);
  2. performs the same string.Template substitution the reference performs
     (same keyword set as aggregation_zeropad.py:131-139);
  3. prepends a ~20-line "CUDA on CPU" shim (blockIdx/threadIdx as thread-locals,
     `__global__` defined away) plus a launcher that walks the exact launch
     geometry the reference uses (block=1024, grid=ceil(n/1024),
     aggregation_zeropad.py:140-141) with OpenMP over blocks;
  4. compiles with g++ into oracle/_ref/<key>.so (git-ignored, but it travels to
     the GPU box with the snapshot) and records it in oracle/_ref/manifest.json.

oracle/_ref is therefore the reference's arithmetic, statement for statement.
It is used (a) to pin oracle/agg_oracle.c and the committed fixtures
(tests/golden/make_golden.py), and (b) as bench.py's `cpu_baseline` with
kind="reference".  On the GPU box /root/reference does not exist: only the
prebuilt .so files listed in the manifest are used there.
"""
import ctypes
import hashlib
import json
import os
import subprocess
import sys
import types
from string import Template

_HERE = os.path.dirname(os.path.abspath(__file__))
REF_DIR = os.path.join(_HERE, "_ref")
MANIFEST = os.path.join(REF_DIR, "manifest.json")
REFERENCE = os.environ.get("COT_REFERENCE", "/root/reference")
CUDA_NUM_THREADS = 1024  # aggregation_zeropad.py:8

_SHIM = r"""
// ---- CUDA-on-CPU shim (oracle/build_ref.py); the kernel text below is the reference's ----
#include <cmath>
struct dim3_ { int x, y, z; };
static thread_local dim3_ blockIdx, threadIdx;
static dim3_ blockDim, gridDim;
#define __global__
"""

_LAUNCH = r"""
extern "C" void launch_${name}(int grid, int block, ${params}) {
  blockDim = {block, 1, 1};
  gridDim = {grid, 1, 1};
  #pragma omp parallel for schedule(static)
  for (int blk_ = 0; blk_ < grid; ++blk_) {
    blockIdx = {blk_, 0, 0};
    for (int thr_ = 0; thr_ < block; ++thr_) {
      threadIdx = {thr_, 0, 0};
      ${name}(${args});
    }
  }
}
"""


def reference_available():
    return os.path.isdir(os.path.join(REFERENCE, "cupy_layers"))


def install_stubs():
    """Stub the two absent third-party modules so reference files import on CPU."""
    if "cupy" not in sys.modules:
        cupy = types.ModuleType("cupy")

        def memoize(for_each_device=False):
            return lambda f: f

        cupy.memoize = memoize
        cupy.cuda = types.SimpleNamespace(compile_with_cache=None)
        sys.modules["cupy"] = cupy
    if "yacs" not in sys.modules:
        yacs, yc = types.ModuleType("yacs"), types.ModuleType("yacs.config")

        class CfgNode(dict):
            def __getattr__(self, k):
                try:
                    return self[k]
                except KeyError:
                    raise AttributeError(k)

            def __setattr__(self, k, v):
                self[k] = v

            def clone(self):
                return self

            def freeze(self):
                pass

            def defrost(self):
                pass

        yc.CfgNode = CfgNode
        yacs.config = yc
        sys.modules["yacs"], sys.modules["yacs.config"] = yacs, yc
    if REFERENCE not in sys.path:
        sys.path.insert(0, REFERENCE)


def _sources():
    install_stubs()
    import cupy_layers.aggregation_zeropad as az
    import cupy_layers.aggregation_zeropad_mix as azm
    return {
        "aggregation_zeropad_forward_kernel": az._aggregation_zeropad_forward_kernel,
        "aggregation_zeropad_input_backward_kernel": az._aggregation_zeropad_input_backward_kernel,
        "aggregation_zeropad_weight_backward_kernel": az._aggregation_zeropad_weight_backward_kernel,
        "aggregation_zeropad_mix_forward_kernel": azm._aggregation_zeropad_mix_forward_kernel,
        "aggregation_zeropad_mix_input_backward_kernel": azm._aggregation_zeropad_mix_input_backward_kernel,
        "aggregation_zeropad_mix_weight_backward_kernel": azm._aggregation_zeropad_mix_weight_backward_kernel,
    }


def _pair(v):
    return (int(v), int(v)) if isinstance(v, int) else (int(v[0]), int(v[1]))


def _out(n, k, s, p, d):
    return int((n + 2 * p - (d * (k - 1) + 1)) / s + 1)


def _key(kind, dtype, **kw):
    blob = json.dumps(dict(kind=kind, dtype=dtype, **kw), sort_keys=True)
    return f"{kind}_{dtype}_" + hashlib.sha1(blob.encode()).hexdigest()[:12], blob


def _load_manifest():
    if os.path.exists(MANIFEST):
        with open(MANIFEST) as f:
            return json.load(f)
    return {}


def _save_manifest(m):
    os.makedirs(REF_DIR, exist_ok=True)
    with open(MANIFEST, "w") as f:
        json.dump(m, f, indent=1, sort_keys=True)


def _compile(key, code):
    os.makedirs(REF_DIR, exist_ok=True)
    src = os.path.join(REF_DIR, key + ".cpp")
    so = os.path.join(REF_DIR, key + ".so")
    with open(src, "w") as f:
        f.write(code)
    subprocess.check_call(["g++", "-O2", "-fPIC", "-shared", "-fopenmp", "-w", "-ffp-contract=off",
                           "-o", so, src])
    if not os.environ.get("COT_KEEP_REF_CPP"):
        os.remove(src)  # the substituted kernel text is the reference's: never leave it lying in the tree
    return so


def _ensure(kind, dtype, names, subst_of, sig_of, **geom):
    """Return path of the .so for this (kind, dtype, geometry); build it if the reference is here."""
    key, blob = _key(kind, dtype, **geom)
    so = os.path.join(REF_DIR, key + ".so")
    man = _load_manifest()
    if os.path.exists(so) and key in man:
        return so
    if not reference_available():
        raise FileNotFoundError(
            f"oracle/_ref has no prebuilt kernel for {blob} and the reference checkout is not present")
    srcs = _sources()
    code = _SHIM
    for nm in names:
        code += Template(srcs[nm]).substitute(**subst_of(nm))
        params, args = sig_of(nm)
        code += Template(_LAUNCH).substitute(name=nm, params=params, args=args)
    _compile(key, code)
    man[key] = json.loads(blob)
    _save_manifest(man)
    return so


class RefAggregation:
    """The reference's three aggregation_zeropad kernels for ONE baked-in geometry, on CPU tensors."""

    def __init__(self, dtype, N, C, H, W, heads, wC, kernel_size, stride, padding, dilation):
        import torch
        self.torch = torch
        k, s, p, d = _pair(kernel_size), _pair(stride), _pair(padding), _pair(dilation)
        self.Ho, self.Wo = _out(H, k[0], s[0], p[0], d[0]), _out(W, k[1], s[1], p[1], d[1])
        self.geom = dict(N=N, C=C, H=H, W=W, heads=heads, wC=wC, k=k, s=s, p=p, d=d)
        self.dtype = dtype
        ctype = {"float": "float", "double": "double"}[dtype]
        n_fwd = N * heads * C * self.Ho * self.Wo          # aggregation_zeropad.py:124
        n_gin = N * C * H * W                              # :170
        n_gw = N * heads * wC * self.Ho * self.Wo          # :179
        self.nthreads = {
            "aggregation_zeropad_forward_kernel": n_fwd,
            "aggregation_zeropad_input_backward_kernel": n_gin,
            "aggregation_zeropad_weight_backward_kernel": n_gw,
        }
        common = dict(Dtype=ctype, num=N, input_channels=C, weight_heads=heads, weight_channels=wC,
                      bottom_height=H, bottom_width=W, top_height=self.Ho, top_width=self.Wo,
                      kernel_h=k[0], kernel_w=k[1], stride_h=s[0], stride_w=s[1],
                      dilation_h=d[0], dilation_w=d[1], pad_h=p[0], pad_w=p[1])

        def subst_of(nm):
            return dict(common, nthreads=self.nthreads[nm])

        def sig_of(nm):
            return (f"const {ctype}* a, const {ctype}* b, {ctype}* c", "a, b, c")

        so = _ensure("agg", dtype, list(self.nthreads), subst_of, sig_of, **self.geom)
        self.lib = ctypes.CDLL(so)

    def _tdtype(self):
        return self.torch.float32 if self.dtype == "float" else self.torch.float64

    def _launch(self, nm, a, b, out):
        n = self.nthreads[nm]
        grid = (n + CUDA_NUM_THREADS - 1) // CUDA_NUM_THREADS
        for t in (a, b, out):
            assert t.is_contiguous() and t.dtype == self._tdtype() and t.device.type == "cpu"
        getattr(self.lib, "launch_" + nm)(ctypes.c_int(grid), ctypes.c_int(CUDA_NUM_THREADS),
                                          ctypes.c_void_p(a.data_ptr()), ctypes.c_void_p(b.data_ptr()),
                                          ctypes.c_void_p(out.data_ptr()))
        return out

    def forward(self, x, w):
        g = self.geom
        out = self.torch.empty(g["N"], g["heads"] * g["C"], self.Ho, self.Wo, dtype=self._tdtype())
        return self._launch("aggregation_zeropad_forward_kernel", x.contiguous(), w.contiguous(), out)

    def backward_input(self, gout, w):
        g = self.geom
        gx = self.torch.empty(g["N"], g["C"], g["H"], g["W"], dtype=self._tdtype())
        return self._launch("aggregation_zeropad_input_backward_kernel", gout.contiguous(), w.contiguous(), gx)

    def backward_weight(self, gout, x):
        g = self.geom
        gw = self.torch.empty(g["N"], g["heads"], g["wC"], g["k"][0] * g["k"][1], self.Ho, self.Wo,
                              dtype=self._tdtype())
        return self._launch("aggregation_zeropad_weight_backward_kernel", gout.contiguous(), x.contiguous(), gw)


class RefAggregationMix:
    """The reference's three aggregation_zeropad_mix kernels (3x3 + 5x5) for one geometry."""

    def __init__(self, dtype, N, C, H, W, heads, wC, stride, padding1, padding2, dilation):
        import torch
        self.torch = torch
        s, p1, p2, d = _pair(stride), _pair(padding1), _pair(padding2), _pair(dilation)
        self.Ho, self.Wo = _out(H, 3, s[0], p1[0], d[0]), _out(W, 3, s[1], p1[1], d[1])  # mix.py:216-217
        self.geom = dict(N=N, C=C, H=H, W=W, heads=heads, wC=wC, s=s, p1=p1, p2=p2, d=d)
        self.dtype = dtype
        ctype = dtype
        n_fwd = N * 2 * heads * C * self.Ho * self.Wo                  # mix.py:221
        n_gin = N * C * H * W                                          # mix.py:273
        n_gw1 = N * heads * wC * self.Ho * self.Wo                     # mix.py:281 (n), nthreads = 2n (:282)
        self.nthreads = {
            "aggregation_zeropad_mix_forward_kernel": n_fwd,
            "aggregation_zeropad_mix_input_backward_kernel": n_gin,
            "aggregation_zeropad_mix_weight_backward_kernel": 2 * n_gw1,
        }
        # the reference launches weight-backward with grid=GET_BLOCKS(n), nthreads=2n (mix.py:283-287)
        self.grid_n = dict(self.nthreads)
        self.grid_n["aggregation_zeropad_mix_weight_backward_kernel"] = n_gw1
        common = dict(Dtype=ctype, num=N, input_channels=C, weight_heads=heads, weight_channels=wC,
                      bottom_height=H, bottom_width=W, top_height=self.Ho, top_width=self.Wo,
                      kernel1_h=3, kernel1_w=3, kernel2_h=5, kernel2_w=5,
                      stride_h=s[0], stride_w=s[1], dilation_h=d[0], dilation_w=d[1],
                      pad1_h=p1[0], pad1_w=p1[1], pad2_h=p2[0], pad2_w=p2[1])

        def subst_of(nm):
            return dict(common, nthreads=self.nthreads[nm])

        def sig_of(nm):
            if nm.endswith("weight_backward_kernel"):
                return (f"const {ctype}* a, const {ctype}* b, {ctype}* c, {ctype}* e", "a, b, c, e")
            return (f"const {ctype}* a, const {ctype}* b, const {ctype}* c, {ctype}* e", "a, b, c, e")

        so = _ensure("mix", dtype, list(self.nthreads), subst_of, sig_of, **self.geom)
        self.lib = ctypes.CDLL(so)

    def _tdtype(self):
        return self.torch.float32 if self.dtype == "float" else self.torch.float64

    def _launch(self, nm, *tensors):
        n = self.grid_n[nm]
        grid = (n + CUDA_NUM_THREADS - 1) // CUDA_NUM_THREADS
        for t in tensors:
            assert t.is_contiguous() and t.dtype == self._tdtype() and t.device.type == "cpu"
        getattr(self.lib, "launch_" + nm)(ctypes.c_int(grid), ctypes.c_int(CUDA_NUM_THREADS),
                                          *[ctypes.c_void_p(t.data_ptr()) for t in tensors])

    def forward(self, x, w1, w2):
        g = self.geom
        out = self.torch.empty(g["N"], 2 * g["heads"] * g["C"], self.Ho, self.Wo, dtype=self._tdtype())
        self._launch("aggregation_zeropad_mix_forward_kernel", x.contiguous(), w1.contiguous(), w2.contiguous(), out)
        return out

    def backward_input(self, gout, w1, w2):
        g = self.geom
        gx = self.torch.empty(g["N"], g["C"], g["H"], g["W"], dtype=self._tdtype())
        self._launch("aggregation_zeropad_mix_input_backward_kernel", gout.contiguous(), w1.contiguous(),
                     w2.contiguous(), gx)
        return gx

    def backward_weight(self, gout, x):
        g = self.geom
        gw1 = self.torch.empty(g["N"], g["heads"], g["wC"], 9, self.Ho, self.Wo, dtype=self._tdtype())
        gw2 = self.torch.empty(g["N"], g["heads"], g["wC"], 25, self.Ho, self.Wo, dtype=self._tdtype())
        self._launch("aggregation_zeropad_mix_weight_backward_kernel", gout.contiguous(), x.contiguous(), gw1, gw2)
        return gw1, gw2


# Geometries prebuilt by `python oracle/build_ref.py` (and by __graft_entry__.build()) so that the
# GPU box -- which has no reference checkout -- finds them in oracle/_ref/.
PREBUILT_AGG = [
    # reference self-tests (aggregation_zeropad.py:238-292)
    dict(dtype="double", N=2, C=8, H=9, W=9, heads=2, wC=4, kernel_size=5, stride=1, padding=2, dilation=1),
    dict(dtype="double", N=2, C=8, H=9, W=9, heads=2, wC=4, kernel_size=1, stride=1, padding=0, dilation=1),
    # BASELINE.json configs[0]: B=2 C=64 H=W=32 k=3
    dict(dtype="float", N=2, C=64, H=32, W=32, heads=1, wC=8, kernel_size=3, stride=1, padding=1, dilation=1),
    dict(dtype="double", N=2, C=64, H=32, W=32, heads=1, wC=8, kernel_size=3, stride=1, padding=1, dilation=1),
    # stride / dilation coverage the reference kernels support but never test
    dict(dtype="double", N=1, C=8, H=11, W=10, heads=2, wC=2, kernel_size=3, stride=2, padding=1, dilation=1),
    dict(dtype="double", N=1, C=8, H=11, W=10, heads=1, wC=4, kernel_size=3, stride=1, padding=2, dilation=2),
    dict(dtype="double", N=1, C=6, H=9, W=12, heads=1, wC=3, kernel_size=(3, 5), stride=(2, 1), padding=(1, 2),
         dilation=(1, 1)),
    # bench.py cpu_baseline sample: the four CoTNet-50 CoT-layer geometries, 4 images, fp32
    dict(dtype="float", N=4, C=64, H=56, W=56, heads=1, wC=8, kernel_size=3, stride=1, padding=1, dilation=1),
    dict(dtype="float", N=4, C=128, H=28, W=28, heads=1, wC=16, kernel_size=3, stride=1, padding=1, dilation=1),
    dict(dtype="float", N=4, C=256, H=14, W=14, heads=1, wC=32, kernel_size=3, stride=1, padding=1, dilation=1),
    dict(dtype="float", N=4, C=512, H=7, W=7, heads=1, wC=64, kernel_size=3, stride=1, padding=1, dilation=1),
    # the same at 32 images per geometry (SURVEY 8d asks for >= 32: 4 images starve the host's OpenMP threads)
    dict(dtype="float", N=32, C=64, H=56, W=56, heads=1, wC=8, kernel_size=3, stride=1, padding=1, dilation=1),
    dict(dtype="float", N=32, C=128, H=28, W=28, heads=1, wC=16, kernel_size=3, stride=1, padding=1, dilation=1),
    dict(dtype="float", N=32, C=256, H=14, W=14, heads=1, wC=32, kernel_size=3, stride=1, padding=1, dilation=1),
    dict(dtype="float", N=32, C=512, H=7, W=7, heads=1, wC=64, kernel_size=3, stride=1, padding=1, dilation=1),
]
PREBUILT_MIX = [
    # reference self-test (aggregation_zeropad_mix.py:344-383)
    dict(dtype="double", N=2, C=8, H=6, W=6, heads=1, wC=4, stride=1, padding1=1, padding2=2, dilation=1),
    dict(dtype="double", N=2, C=8, H=6, W=6, heads=2, wC=4, stride=1, padding1=1, padding2=2, dilation=1),
]


def build_all(verbose=True):
    if not reference_available():
        if verbose:
            print(f"[build_ref] {REFERENCE} not present: keeping prebuilt oracle/_ref as is")
        return False
    for g in PREBUILT_AGG:
        RefAggregation(**g)
    for g in PREBUILT_MIX:
        RefAggregationMix(**g)
    if verbose:
        print(f"[build_ref] {len(_load_manifest())} reference kernels available in {REF_DIR}")
    return True


if __name__ == "__main__":
    build_all()

"""The randomised differential cases of tests/test_fuzz_emulated.py against the DEVICE library on the MI355X: the same case functions (random
batch sizes, channel counts, plane sizes from 1 x 1 up, slabs, biases, forced split counts, tuning keys, operands inside NaN margins), with the
tensors on `cuda:0` and `libcotnet_hip.so` in the emulator's place -- the shapes nobody writes down, through the real MFMA / LDS-DMA / buffer-load
paths instead of their host model.  References: torch on the device (fp32 / fp64) and the C oracle (oracle/cref.py, on host copies).
`scripts/fuzz_gpu.py` runs the same loop for as many seeds as one likes (profiles/r06_fuzz_gpu.log)."""
import contextlib
import random

import pytest
import torch

pytestmark = pytest.mark.gpu


class _DeviceLib:
    """libcotnet_hip.so with the emulator-only entry points as no-ops"""

    _test_device = "cuda"  # (tests/bn_tail_cases.py: where the shared cases put their tensors)

    def __init__(self, lib):
        self._lib = lib

    def __getattr__(self, name):
        if name == "emul_set_dma_mode":  # (host model of the two landing times of an LDS copy: the hardware has its own)
            return lambda mode: 0
        return getattr(self._lib, name)


def _on_host(fn):
    def wrapped(*args, **kw):
        torch.set_default_device("cpu")  # (the oracle allocates its outputs itself)
        try:
            out = fn(*[a.cpu() if isinstance(a, torch.Tensor) else a for a in args], **kw)
        finally:
            torch.set_default_device("cuda")
        if isinstance(out, torch.Tensor):
            return out.cuda()
        return tuple(o.cuda() if isinstance(o, torch.Tensor) else o for o in out)
    return wrapped


@contextlib.contextmanager
def device_fuzz():
    """the emulated fuzz module re-pointed at the device: library, default tensor device, the oracle behind host copies"""
    from cotnet_amd import _lib
    from oracle import cref
    from tests import test_fuzz_emulated as tfe, test_kernels_emulated as tke
    dev = _DeviceLib(_lib.lib())  # (raises when the HIP library is missing: no fallback)
    saved = (tfe.E, tke._EMUL, tke.oracle_all)
    tfe.ON_DEVICE = True
    oracle_fns = {n: getattr(cref, n) for n in ("mix_forward", "mix_backward_input", "mix_backward_weight")}
    tfe.E, tke._EMUL, tke.oracle_all = dev, dev, _on_host(tke.oracle_all)
    for n, f in oracle_fns.items():
        setattr(cref, n, _on_host(f))
    torch.set_default_device("cuda")
    try:
        yield tfe
    finally:
        torch.set_default_device("cpu")
        tfe.E, tke._EMUL, tke.oracle_all = saved
        tfe.ON_DEVICE = False
        for n, f in oracle_fns.items():
            setattr(cref, n, f)
        for k, v in ((10, 0), (11, 2048)):
            dev.cot_set_tuning(k, v)


def run_cases(tfe, seed, count, cases=None):
    rng = random.Random(seed)
    torch.manual_seed(seed)
    cases = cases or (tfe.CASES + tfe.CASES_R4 + tfe.CASES_R6 + tfe.CASES_R6B)
    failures, kinds = [], {}
    for _ in range(count):
        fn = rng.choice(cases)
        ok, desc = fn(rng)
        torch.cuda.synchronize()
        kinds[fn.__name__] = kinds.get(fn.__name__, 0) + 1
        if not ok:
            failures.append(desc)
    return failures, kinds


@pytest.mark.parametrize("seed", [601, 602, 603, 604])
def test_random_shapes_on_the_device(seed):
    with device_fuzz() as tfe:
        failures, _ = run_cases(tfe, seed, 60)
    assert not failures, failures

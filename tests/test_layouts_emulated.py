"""The per-tensor-layout kernels (cot_*_lay) on the host emulator: tests/layout_cases.py with CPU tensors."""
import ctypes

import pytest

from cotnet_amd import _lib
from tests import layout_cases as lc
from tests.emul import build_emul

try:
    _EMUL = ctypes.CDLL(build_emul.build())
    for _name, (_res, _args) in _lib.SYMBOLS.items():
        getattr(_EMUL, _name).restype = _res
        getattr(_EMUL, _name).argtypes = _args
except FileNotFoundError:
    _EMUL = None

pytestmark = pytest.mark.skipif(_EMUL is None, reason="host emulation build unavailable")


@pytest.fixture(autouse=True)
def _knobs():
    _EMUL.cot_set_tuning(18, 256)
    _EMUL.cot_set_tuning(21, 1)  # channel-resident BatchNorm kernels on
    yield


@pytest.mark.parametrize("HW", [196, 49, 64])
@pytest.mark.parametrize("act,res,y2,ps", [(1, False, True, False), (0, False, False, False), (2, False, False, False), (1, True, False, True)])
def test_batchnorm_forward_layouts(HW, act, res, y2, ps):
    lc.bn_forward_case(_EMUL, "cpu", None, 6, 8, HW, act, res, y2, ps)


@pytest.mark.parametrize("HW", [196, 49, 64])
@pytest.mark.parametrize("act,res,dy2,ps", [(1, False, True, False), (0, False, False, False), (2, False, False, False), (1, True, False, True)])
def test_batchnorm_backward_layouts(HW, act, res, dy2, ps):
    lc.bn_backward_case(_EMUL, "cpu", None, 6, 8, HW, act, res, dy2, ps)


@pytest.mark.parametrize("HW", [196, 49])
def test_radix_tail_layouts(HW):
    lc.radix_case(_EMUL, "cpu", None, 5, 16, HW)


@pytest.mark.parametrize("HW", [196, 49, 784])
def test_group_norm9_layouts(HW):
    lc.gn9_case(_EMUL, "cpu", None, 3, 4, HW)

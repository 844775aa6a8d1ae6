"""The wrappers' fallbacks to the torch modules are counted, and COT_STRICT_DISPATCH=1 refuses them (cotnet_amd/_lib.py: fallback)."""
import pytest
import torch

from cotnet_amd import _lib, conv3x3g


class _OnGpu:  # (what the counter looks at: a tensor that lives on a GPU)
    is_cuda, shape, dtype = True, (2, 3, 5, 5), torch.bfloat16


def test_fallbacks_are_counted_per_site_and_cpu_tensors_do_not_count(monkeypatch):
    monkeypatch.setattr(_lib, "FALLBACKS", {})
    monkeypatch.setattr(_lib, "STRICT_DISPATCH", False)
    _lib.fallback("conv3x3", _OnGpu(), "stride (2, 2)")
    _lib.fallback("conv3x3", _OnGpu())
    _lib.fallback("pool", _OnGpu())
    _lib.fallback("pool", torch.zeros(1))  # (a CPU tensor: the modules run torch there by design)
    assert _lib.FALLBACKS == {"conv3x3": 2, "pool": 1}


def test_strict_dispatch_turns_a_fallback_into_an_error(monkeypatch):
    monkeypatch.setattr(_lib, "FALLBACKS", {})
    monkeypatch.setattr(_lib, "STRICT_DISPATCH", True)
    with pytest.raises(RuntimeError, match="COT_STRICT_DISPATCH"):
        _lib.fallback("stem_conv", _OnGpu(), "-> 32")
    _lib.fallback("stem_conv", torch.zeros(1))  # (CPU: never an error)


def test_a_wrapper_reports_the_module_it_falls_back_to(monkeypatch):
    """a strided 3x3 convolution is off the grouped-3x3 kernels' grid: the wrapper runs the module and says so"""
    monkeypatch.setattr(_lib, "FALLBACKS", {})
    monkeypatch.setattr(_lib, "STRICT_DISPATCH", False)
    monkeypatch.setattr(conv3x3g, "MODE", "hip")
    seen = []
    monkeypatch.setattr(_lib, "fallback", lambda site, x=None, detail="": seen.append((site, detail)))
    conv = torch.nn.Conv2d(8, 8, 3, stride=2, padding=1, bias=False)
    y = conv3x3g.conv3x3(conv, torch.randn(1, 8, 6, 6))
    assert y.shape == (1, 8, 3, 3) and seen and seen[0][0] == "conv3x3" and "stride (2, 2)" in seen[0][1]

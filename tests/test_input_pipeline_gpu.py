"""On-device input pipeline (cot_input_normalize / cotnet_amd.input_pipeline.PrefetchLoader) against the reference's
PrefetchLoader arithmetic (datasets/loader.py:85-90: `.float().sub_(mean).div_(std)` and the `.half()` form)."""
import pytest
import torch

from cotnet_amd.input_pipeline import IMAGENET_DEFAULT_MEAN, IMAGENET_DEFAULT_STD, PrefetchLoader, normalize_uint8

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _consts(dtype=torch.float32):
    mean = torch.tensor([v * 255 for v in IMAGENET_DEFAULT_MEAN], device=DEV).view(1, 3, 1, 1)
    std = torch.tensor([v * 255 for v in IMAGENET_DEFAULT_STD], device=DEV).view(1, 3, 1, 1)
    return mean.to(dtype), std.to(dtype)


@pytest.mark.parametrize("shape", [(80, 3, 224, 224), (3, 3, 7, 9), (2, 3, 32, 32), (1, 3, 320, 320)])
def test_fp32_is_bit_identical_to_the_reference_arithmetic(shape):
    g = torch.Generator().manual_seed(1)
    x = torch.randint(0, 256, shape, dtype=torch.uint8, generator=g).to(DEV)
    mean, std = _consts()
    ref = x.float().sub_(mean).div_(std)
    got = normalize_uint8(x, mean.flatten().contiguous(), std.flatten().contiguous(), torch.float32)
    assert torch.equal(got, ref)


def test_fp16_matches_the_reference_half_path_and_bf16_rounds_once():
    g = torch.Generator().manual_seed(2)
    x = torch.randint(0, 256, (4, 3, 64, 48), dtype=torch.uint8, generator=g).to(DEV)
    mh, sh = _consts(torch.float16)
    ref = x.half().sub_(mh).div_(sh)   # the reference's fp16=True branch
    got = normalize_uint8(x, mh.float().flatten().contiguous(), sh.float().flatten().contiguous(), torch.float16)
    assert torch.equal(got, ref)
    mean, std = _consts()
    got_b = normalize_uint8(x, mean.flatten().contiguous(), std.flatten().contiguous(), torch.bfloat16)
    assert torch.equal(got_b, x.float().sub_(mean).div_(std).bfloat16())


def test_prefetch_loader_yields_the_reference_batches_in_order():
    g = torch.Generator().manual_seed(3)
    batches = [(torch.randint(0, 256, (5, 3, 16, 16), dtype=torch.uint8, generator=g), torch.arange(5) + 10 * i)
               for i in range(4)]
    mean, std = _consts()
    out = list(PrefetchLoader(batches))
    assert len(out) == 4 and len(PrefetchLoader(batches)) == 4
    for (xi, ti), (xo, to) in zip(batches, out):
        assert xo.dtype == torch.float32 and xo.is_cuda
        assert torch.equal(xo, xi.to(DEV).float().sub_(mean).div_(std))
        assert torch.equal(to.cpu(), ti)
    out_b = list(PrefetchLoader(batches, dtype=torch.bfloat16))
    assert out_b[2][0].dtype == torch.bfloat16
    with pytest.raises(NotImplementedError):
        PrefetchLoader(batches, re_prob=0.5)

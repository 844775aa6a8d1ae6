"""On-device input pipeline -- drop-in for the reference's `PrefetchLoader` (datasets/loader.py:54-102).

The reference wraps a host DataLoader that yields uint8 NCHW batches (`fast_collate`, loader.py:20-50), copies each batch
to the GPU on a side stream and normalises it there with three elementwise kernels
(`.float()` / `.half()`, `.sub_(mean)`, `.div_(std)`; mean/std = 255 x the ImageNet constants, :66-67), one batch ahead
of the consumer.  Here the three kernels are ONE hand-written pass (`cot_input_normalize`, csrc/input_norm.hip: 1 B read
+ 2/4 B written per element) that writes the model's input dtype directly (fp32, the reference's fp16, or bf16 for the
bf16 model); the side stream, the non-blocking copy and the one-batch look-ahead are kept.  fp32 results are
bit-identical to the reference's (IEEE subtract, then IEEE divide).  Random erasing (`re_prob`) is data augmentation and
out of scope: a non-zero `re_prob` raises.
"""
import ctypes

import torch

from . import _lib

IMAGENET_DEFAULT_MEAN = (0.485, 0.456, 0.406)
IMAGENET_DEFAULT_STD = (0.229, 0.224, 0.225)
_DEVICE_ONLY = True  # tests drive the wrapper on CPU tensors through the host-emulated kernels


def normalize_uint8(x, mean, std, dtype=torch.float32, out=None):
    """x: uint8 [N, C, H, W] on the GPU (contiguous); mean / std: fp32 device tensors of C entries, already scaled by 255
    -> (x - mean[c]) / std[c] as `dtype`, one kernel on the current stream"""
    if x.dtype != torch.uint8 or x.dim() != 4 or not x.is_contiguous():
        raise TypeError("normalize_uint8: expects a contiguous uint8 NCHW tensor")
    if _DEVICE_ONLY and not x.is_cuda:
        raise RuntimeError("normalize_uint8: cotnet_amd has no CPU path (input must be on the GPU)")
    N, C, H, W = x.shape
    if mean.numel() != C or std.numel() != C or mean.dtype != torch.float32 or std.dtype != torch.float32:
        raise ValueError("normalize_uint8: mean / std must be fp32 tensors with one entry per channel")
    y = torch.empty((N, C, H, W), dtype=dtype, device=x.device) if out is None else out
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream) if _DEVICE_ONLY else None
    rc = _lib.lib().cot_input_normalize(x.data_ptr(), y.data_ptr(), mean.data_ptr(), std.data_ptr(), N * C, C, H * W,
                                        _lib.dtype_code(dtype), stream)
    _lib.check(rc, "cot_input_normalize")
    return y


class PrefetchLoader:
    """Same constructor and iteration protocol as the reference's (loader.py:56-102); `dtype` is the one addition
    (default: fp16 if `fp16` else fp32, as the reference; torch.bfloat16 feeds the bf16 model without a cast kernel)."""

    def __init__(self, loader, mean=IMAGENET_DEFAULT_MEAN, std=IMAGENET_DEFAULT_STD, fp16=False, re_prob=0.0,
                 re_mode="const", re_count=1, re_num_splits=0, dtype=None, device="cuda"):
        if re_prob > 0.0:
            raise NotImplementedError("RandomErasing is data augmentation: out of scope (SURVEY.md 2)")
        self.loader = loader
        self.dtype = dtype if dtype is not None else (torch.float16 if fp16 else torch.float32)
        self.device = torch.device(device)
        mean = torch.tensor([v * 255 for v in mean], dtype=torch.float32)
        std = torch.tensor([v * 255 for v in std], dtype=torch.float32)
        if self.dtype == torch.float16:  # the reference rounds the constants to half first (loader.py:69-71)
            mean, std = mean.half().float(), std.half().float()
        self.mean, self.std = mean.to(self.device), std.to(self.device)

    def __iter__(self):
        stream = torch.cuda.Stream(device=self.device)
        first = True
        inp = tgt = None
        for next_input, next_target in self.loader:
            with torch.cuda.stream(stream):
                next_input = next_input.to(self.device, non_blocking=True)
                next_target = next_target.to(self.device, non_blocking=True)
                next_input = normalize_uint8(next_input.contiguous(), self.mean, self.std, self.dtype)
            if not first:
                yield inp, tgt
            else:
                first = False
            torch.cuda.current_stream(self.device).wait_stream(stream)
            # the batch was produced on the side stream and is consumed on the current one: tell the allocator
            next_input.record_stream(torch.cuda.current_stream(self.device))
            next_target.record_stream(torch.cuda.current_stream(self.device))
            inp, tgt = next_input, next_target
        if inp is not None:
            yield inp, tgt

    def __len__(self):
        return len(self.loader)

    @property
    def sampler(self):
        return self.loader.sampler

    @property
    def dataset(self):
        return self.loader.dataset

"""The handful of timm layers the CoT models touch (reference: models/layers/*).

Only what cotnet.py / cotnet_hybrid.py / the ResNet skeleton instantiate is provided; parameter-bearing
sub-module names match the reference so checkpoints load with strict=True:
  get_act_layer('swish') -> nn.SiLU          models/layers/create_act.py:52-53,:107-120
  DropPath                                    models/layers/drop.py:140-168
  SelectAdaptivePool2d / create_classifier    models/layers/adaptive_avgmax_pool.py:79, classifier.py:11-25
  AvgPool2dSame                               models/layers/pool2d_same.py:14-31
  BlurPool2d                                  models/layers/blur_pool.py:18-58
  SplitAttnConv2d (+ RadixSoftmax)            models/layers/split_attn.py:14-88
"""
import math

import torch
import torch.nn.functional as F
from torch import nn

from . import se_gate
from .conv3x3g import conv3x3
from .fused_bn import fused_bn_act
from .se_gate import se_mlp


def get_act_layer(name="relu"):
    table = {"relu": nn.ReLU, "swish": nn.SiLU, "silu": nn.SiLU, "sigmoid": nn.Sigmoid, "gelu": nn.GELU}
    if name is None:
        return None
    if name not in table:
        raise KeyError(f"activation '{name}' is not used by the CoT models and is not provided")
    return table[name]


def drop_path(x, drop_prob: float = 0.0, training: bool = False):
    if drop_prob == 0.0 or not training:
        return x
    keep = 1.0 - drop_prob
    mask = x.new_empty((x.shape[0],) + (1,) * (x.ndim - 1)).uniform_().add_(keep).floor_()
    return x.div(keep) * mask


class DropPath(nn.Module):
    def __init__(self, drop_prob=None):
        super().__init__()
        self.drop_prob = drop_prob

    def forward(self, x):
        return drop_path(x, self.drop_prob, self.training)


class SelectAdaptivePool2d(nn.Module):
    """global pooling head; only the pool types reachable from the CoT entry points"""

    def __init__(self, output_size=1, pool_type="avg", flatten=False):
        super().__init__()
        self.pool_type = pool_type or ""
        self.flatten = flatten
        if self.pool_type == "":
            self.pool = nn.Identity()
        elif self.pool_type == "avg":
            self.pool = nn.AdaptiveAvgPool2d(output_size)
        elif self.pool_type == "max":
            self.pool = nn.AdaptiveMaxPool2d(output_size)
        else:
            raise ValueError(f"pool type '{pool_type}' not provided (avg / max / '')")

    def is_identity(self):
        return self.pool_type == ""

    def feat_mult(self):
        return 1

    def forward(self, x):
        x = self.pool(x)
        return x.flatten(1) if self.flatten else x


def create_classifier(num_features, num_classes, pool_type="avg", use_conv=False):
    flatten = not use_conv
    if not pool_type:
        assert num_classes == 0 or use_conv
        flatten = False
    global_pool = SelectAdaptivePool2d(pool_type=pool_type, flatten=flatten)
    feats = num_features * global_pool.feat_mult()
    if num_classes <= 0:
        fc = nn.Identity()
    elif use_conv:
        fc = nn.Conv2d(feats, num_classes, 1, bias=True)
    else:
        fc = nn.Linear(feats, num_classes, bias=True)
    return global_pool, fc


def _same_pad(x, k, s, d=1):
    return max((math.ceil(x / s) - 1) * s + (k - 1) * d + 1 - x, 0)


class AvgPool2dSame(nn.AvgPool2d):
    """TF-'SAME' average pooling: pad dynamically, then pool without padding"""

    def __init__(self, kernel_size, stride=None, padding=0, ceil_mode=False, count_include_pad=True):
        ks = kernel_size if isinstance(kernel_size, (tuple, list)) else (kernel_size, kernel_size)
        st = stride if isinstance(stride, (tuple, list)) else (stride, stride)
        super().__init__(tuple(ks), tuple(st), (0, 0), ceil_mode, count_include_pad)

    def forward(self, x):
        ih, iw = x.shape[-2:]
        ph = _same_pad(ih, self.kernel_size[0], self.stride[0])
        pw = _same_pad(iw, self.kernel_size[1], self.stride[1])
        if ph > 0 or pw > 0:
            x = F.pad(x, [pw // 2, pw - pw // 2, ph // 2, ph - ph // 2])
        return F.avg_pool2d(x, self.kernel_size, self.stride, (0, 0), self.ceil_mode, self.count_include_pad)


class BlurPool2d(nn.Module):
    """anti-aliased down-sampling: reflection pad + depthwise binomial filter with stride (no parameters,
    no buffers -- the filter is rebuilt per device/dtype like the reference's lazy cache)"""

    def __init__(self, channels, filt_size=3, stride=2):
        super().__init__()
        assert filt_size > 1
        self.channels = channels
        self.filt_size = filt_size
        self.stride = stride
        pad = ((stride - 1) + (filt_size - 1)) // 2
        self.padding = nn.ReflectionPad2d([pad] * 4)
        row = torch.tensor([math.comb(filt_size - 1, i) for i in range(filt_size)], dtype=torch.float64)
        self._coeffs = row / row.sum()
        self._cache = {}

    def _apply(self, fn):
        self._cache = {}
        return super()._apply(fn)

    def _filter(self, like):
        key = (str(like.device), like.dtype)
        f = self._cache.get(key)
        if f is None:
            k2 = (self._coeffs[:, None] * self._coeffs[None, :]).to(dtype=like.dtype, device=like.device)
            f = k2[None, None].repeat(self.channels, 1, 1, 1)
            self._cache[key] = f
        return f

    def forward(self, x):
        from . import pool3x3  # (late: pool3x3 imports nothing from here, but keep module import order flat)
        if pool3x3.MODE == "hip" and self.filt_size == 3 and self.stride == 2 and pool3x3.blur_eligible(x):
            return pool3x3.blur_pool(x)  # the same 9-tap stencil on csrc/pool3x3.hip (reflection folded into the indices)
        return F.conv2d(self.padding(x), self._filter(x), stride=self.stride, groups=x.shape[1])


class RadixSoftmax(nn.Module):
    def __init__(self, radix, cardinality):
        super().__init__()
        self.radix = radix
        self.cardinality = cardinality

    def forward(self, x):
        b = x.size(0)
        if self.radix > 1:
            x = x.view(b, self.cardinality, self.radix, -1).transpose(1, 2)
            return F.softmax(x, dim=1).reshape(b, -1)
        return torch.sigmoid(x)


class SplitAttnConv2d(nn.Module):
    """ResNeSt split-attention conv; SE-CoTNetD uses it with radix=1 (i.e. conv+bn+act followed by an
    SE-style sigmoid gate), models/cotnet_hybrid.py:143-146."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1, bias=False,
                 radix=2, reduction_factor=4, act_layer=nn.ReLU, norm_layer=None, drop_block=None, **kwargs):
        super().__init__()
        self.radix = radix
        self.drop_block = drop_block
        mid = out_channels * radix
        attn = max(in_channels * radix // reduction_factor, 32)
        self.conv = nn.Conv2d(in_channels, mid, kernel_size, stride, padding, dilation, groups=groups * radix,
                              bias=bias, **kwargs)
        self.bn0 = norm_layer(mid) if norm_layer is not None else None
        self.act0 = act_layer(inplace=True)
        self.fc1 = nn.Conv2d(out_channels, attn, 1, groups=groups)
        self.bn1 = norm_layer(attn) if norm_layer is not None else None
        self.act1 = act_layer(inplace=True)
        self.fc2 = nn.Conv2d(attn, mid, 1, groups=groups)
        self.rsoftmax = RadixSoftmax(radix, groups)

    @property
    def in_channels(self):
        return self.conv.in_channels

    @property
    def out_channels(self):
        return self.fc1.out_channels

    def forward(self, x):
        x = conv3x3(self.conv, x)  # the module itself unless COT_CONV3X3=hip and the tensor qualifies
        a0 = "relu" if isinstance(self.act0, nn.ReLU) else ("silu" if isinstance(self.act0, nn.SiLU) else None)
        if isinstance(self.bn0, nn.BatchNorm2d) and self.drop_block is None and a0 is not None:
            # one fused op when eligible, the same modules otherwise (SE-CoTNetD passes act_layer = swish: models/cotnet_hybrid.py:143-146)
            x = fused_bn_act(x, self.bn0, a0)
        else:
            if self.bn0 is not None:
                x = self.bn0(x)
            if self.drop_block is not None:
                x = self.drop_block(x)
            x = self.act0(x)
        B, RC, H, W = x.shape
        if self.radix == 1 and se_gate.eligible(x) and isinstance(self.bn1, nn.BatchNorm2d) and self.fc1.groups == 1:
            # SE-CoTNetD's form: pooled descriptor -> fc1 / bn1 / act1 / fc2 on [B, C] -> sigmoid gate, three passes over x
            logits = se_mlp(se_gate.se_gap(x), (self.fc1, self.bn1, self.act1, self.fc2))
            return se_gate.se_gate(x, logits)
        if self.radix > 1:
            x = x.reshape(B, self.radix, RC // self.radix, H, W)
            gap = x.sum(dim=1)
        else:
            gap = x
        gap = self.fc1(F.adaptive_avg_pool2d(gap, 1))
        if self.bn1 is not None:
            gap = self.bn1(gap)
        attn = self.rsoftmax(self.fc2(self.act1(gap))).view(B, -1, 1, 1)
        if self.radix > 1:
            out = (x * attn.reshape(B, self.radix, RC // self.radix, 1, 1)).sum(dim=1)
        else:
            out = x * attn
        return out.contiguous()

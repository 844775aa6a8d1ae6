"""CoTNet / CoTNeXt on MI355X -- drop-in for the reference's models/cotnet.py.

`CotLayer` (ref :36-104), `CoXtLayer` (ref :106-178), `Bottleneck` (ref :181-264) and the entry points
`cotnet50`, `cotnext50_2x48d`, `cotnet101`, `cotnext101_2x48d` (ref :270-288) keep the reference's constructor
signatures, sub-module names and parameter shapes, so a reference checkpoint loads with
strict=True.  The forward pass computes the same function; what differs is how:

  * the local aggregation is the hand-written HIP kernel behind `LocalConvolution`
    (cotnet_amd/aggregation_zeropad.py), not CuPy-JIT CUDA;
  * the radix-2 tail (ref :92-104: view/cat to [B,C,2,H,W], sum, mean, softmax, broadcast-multiply, sum)
    never materialises the 5-D tensors: GAP(x + k), then `x*a0 + k*a1`;
  * CoXtLayer's "fold the 2 groups into the batch" views (ref :157-162) are kept -- they are free views in
    NCHW -- so the same aggregation kernel serves CoTNeXt with N'=2B, C'=C/2.
"""
import math

import torch
import torch.nn.functional as F
from torch import nn

from .aggregation_zeropad import LocalConvolution
from . import cot_layer_fused, radix_tail
from .conv1x1 import conv1x1, run_downsample
from .conv3x3g import conv3x3
from .group_norm9 import group_norm9
from .pool3x3 import pool
from .fused_bn import fused_bn_act
from .se_gate import se_mlp  # noqa: F401  (re-exported: the se branch of the CoT layers and SplitAttnConv2d's gate share it)
from .layers import get_act_layer
from .registry import build_model_with_cfg, register_model
from .resnet import ResNet

IMAGENET_DEFAULT_MEAN = (0.485, 0.456, 0.406)
IMAGENET_DEFAULT_STD = (0.229, 0.224, 0.225)


def _cfg(url="", **kwargs):
    return {"url": url, "num_classes": 1000, "input_size": (3, 224, 224), "pool_size": (7, 7), "crop_pct": 0.875,
            "interpolation": "bicubic", "mean": IMAGENET_DEFAULT_MEAN, "std": IMAGENET_DEFAULT_STD,
            "first_conv": "conv1", "classifier": "fc", **kwargs}


default_cfgs = {"cot_basic": _cfg(url="")}


def act_name(m):
    """'relu' / 'silu' / None for the activation modules the fused BN kernels implement, else False"""
    if m is None or isinstance(m, nn.Identity):
        return None
    if isinstance(m, nn.ReLU):
        return "relu"
    if isinstance(m, nn.SiLU):
        return "silu"
    return False


def radix2_fuse(x, k, se):
    """ref :92-104 without the [B,C,2,H,W] temporaries.
    attn = softmax over the radix pair of se(GAP(x + k)), channel index of se's output = c*2 + r (:100)."""
    B, C = x.shape[:2]
    if radix_tail.eligible(x, k):  # two fused HIP ops around the tiny se MLP (csrc/radix_tail.hip)
        attn = F.softmax(se_mlp(radix_tail.radix_gap(x, k), se).view(B, C, 2), dim=2)
        return radix_tail.radix_mix(x, k, attn)
    gap = (x + k).mean((2, 3), keepdim=True)
    attn = F.softmax(se_mlp(gap, se).view(B, C, 2), dim=2)
    a0 = attn[:, :, 0].reshape(B, C, 1, 1)
    a1 = attn[:, :, 1].reshape(B, C, 1, 1)
    return (x * a0 + k * a1).contiguous()


class CotLayer(nn.Module):
    def __init__(self, dim, kernel_size):
        super(CotLayer, self).__init__()
        self.dim = dim
        self.kernel_size = kernel_size

        # static context: 3x3 grouped conv over the keys (ref :43-47)
        self.key_embed = nn.Sequential(
            nn.Conv2d(dim, dim, self.kernel_size, stride=1, padding=self.kernel_size // 2, groups=4, bias=False),
            nn.BatchNorm2d(dim),
            nn.ReLU(inplace=True))

        share_planes = 8
        factor = 2
        # [query, key] -> k*k attention logits per group of 8 channels (ref :51-57)
        self.embed = nn.Sequential(
            nn.Conv2d(2 * dim, dim // factor, 1, bias=False),
            nn.BatchNorm2d(dim // factor),
            nn.ReLU(inplace=True),
            nn.Conv2d(dim // factor, pow(kernel_size, 2) * dim // share_planes, kernel_size=1),
            nn.GroupNorm(num_groups=dim // share_planes, num_channels=pow(kernel_size, 2) * dim // share_planes))

        # values (ref :59-62)
        self.conv1x1 = nn.Sequential(
            nn.Conv2d(dim, dim, kernel_size=1, stride=1, padding=0, dilation=1, bias=False),
            nn.BatchNorm2d(dim))

        self.local_conv = LocalConvolution(dim, dim, kernel_size=self.kernel_size, stride=1,
                                           padding=(self.kernel_size - 1) // 2, dilation=1)
        self.bn = nn.BatchNorm2d(dim)
        self.act = get_act_layer("swish")(inplace=True)

        reduction_factor = 4
        self.radix = 2
        attn_chs = max(dim * self.radix // reduction_factor, 32)
        self.se = nn.Sequential(
            nn.Conv2d(dim, attn_chs, 1),
            nn.BatchNorm2d(attn_chs),
            nn.ReLU(inplace=True),
            nn.Conv2d(attn_chs, self.radix * dim, 1))

    def forward(self, x):
        if cot_layer_fused.ENABLED and cot_layer_fused.eligible(self, x):
            return cot_layer_fused.cot_layer_forward(self, x)  # the whole layer as one autograd node (opt-in)
        # Sequential members are called one by one so that each BatchNorm runs fused with its activation
        # (cotnet_amd.fused_bn); module structure and state_dict keys are the reference's.
        k = fused_bn_act(conv3x3(self.key_embed[0], x), self.key_embed[1], "relu")
        b, _, qk_hh, qk_ww = x.size()

        # embed[0] consumes the concatenation [x, k] (ref :81); conv1x1 reads the two slabs in place when it can
        w = fused_bn_act(conv1x1(self.embed[0], x, k), self.embed[1], "relu")
        w = group_norm9(self.embed[4], conv1x1(self.embed[3], w))
        w = w.view(b, 1, -1, self.kernel_size * self.kernel_size, qk_hh, qk_ww)

        x = fused_bn_act(conv1x1(self.conv1x1[0], x), self.conv1x1[1], None)
        x = self.local_conv(x, w)
        x = fused_bn_act(x, self.bn, act_name(self.act) or None) if act_name(self.act) is not False else self.act(self.bn(x))
        return radix2_fuse(x, k, self.se)


class CoXtLayer(nn.Module):
    def __init__(self, dim, kernel_size):
        super(CoXtLayer, self).__init__()
        self.dim = dim
        self.kernel_size = kernel_size

        self.key_embed = nn.Sequential(
            nn.Conv2d(dim, dim, self.kernel_size, stride=1, padding=self.kernel_size // 2, groups=8, bias=False),
            nn.BatchNorm2d(dim),
            nn.ReLU(inplace=True))

        self.dw_group = 2
        share_planes = 8
        factor = 2
        self.embed = nn.Sequential(
            nn.Conv2d(2 * dim, dim // factor, 1, groups=self.dw_group, bias=False),
            nn.BatchNorm2d(dim // factor),
            nn.ReLU(inplace=True),
            nn.Conv2d(dim // factor, pow(kernel_size, 2) * dim // share_planes, kernel_size=1, groups=self.dw_group),
            nn.GroupNorm(num_groups=dim // share_planes, num_channels=pow(kernel_size, 2) * dim // share_planes))

        self.conv1x1 = nn.Sequential(
            nn.Conv2d(dim, dim, kernel_size=1, stride=1, padding=0, dilation=1, groups=self.dw_group, bias=False),
            nn.BatchNorm2d(dim))

        self.local_conv = LocalConvolution(dim, dim, kernel_size=self.kernel_size, stride=1,
                                           padding=(self.kernel_size - 1) // 2, dilation=1)
        self.bn = nn.BatchNorm2d(dim)
        self.act = get_act_layer("swish")(inplace=True)

        reduction_factor = 4
        self.radix = 2
        attn_chs = max(dim * self.radix // reduction_factor, 32)
        self.se = nn.Sequential(
            nn.Conv2d(dim, attn_chs, 1),
            nn.BatchNorm2d(attn_chs),
            nn.ReLU(inplace=True),
            nn.Conv2d(attn_chs, self.radix * dim, 1))

    def forward(self, x):
        if cot_layer_fused.ENABLED and cot_layer_fused.eligible(self, x):
            return cot_layer_fused.cot_layer_forward(self, x)  # the whole layer as one autograd node
        batch_size, channels, height, width = x.size()
        k = fused_bn_act(conv3x3(self.key_embed[0], x), self.key_embed[1], "relu")
        # channel-interleave [x0,k0,x1,k1,...] so each of the 2 conv groups sees matching x/k halves (ref :153-154)
        qk = torch.stack([x, k], dim=2).view(batch_size, -1, height, width)

        w = fused_bn_act(conv1x1(self.embed[0], qk), self.embed[1], "relu")
        w = group_norm9(self.embed[4], conv1x1(self.embed[3], w))
        w = w.reshape(batch_size * self.dw_group, 1, -1, self.kernel_size * self.kernel_size, height, width)

        x = fused_bn_act(conv1x1(self.conv1x1[0], x), self.conv1x1[1], None)
        x = x.reshape(batch_size * self.dw_group, -1, height, width)
        x = self.local_conv(x, w)
        x = x.view(batch_size, -1, height, width)
        x = fused_bn_act(x, self.bn, act_name(self.act) or None) if act_name(self.act) is not False else self.act(self.bn(x))
        return radix2_fuse(x, k, self.se)


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None, cardinality=1, base_width=64, reduce_first=1,
                 dilation=1, first_dilation=None, act_layer=nn.ReLU, norm_layer=nn.BatchNorm2d, attn_layer=None,
                 aa_layer=None, drop_block=None, drop_path=None):
        super(Bottleneck, self).__init__()
        assert attn_layer is None, "attn_layer is unused by every CoT entry point"
        width = int(math.floor(planes * (base_width / 64)) * cardinality)
        first_planes = width // reduce_first
        outplanes = planes * self.expansion

        self.conv1 = nn.Conv2d(inplanes, first_planes, kernel_size=1, bias=False)
        self.bn1 = norm_layer(first_planes)
        self.act1 = act_layer(inplace=True)

        # stride is taken by an average pool BEFORE the CoT layer, which always runs at stride 1 (ref :237-240)
        self.avd = nn.AvgPool2d(3, 2, padding=1) if stride > 1 else None
        self.conv2 = CotLayer(width, kernel_size=3) if cardinality == 1 else CoXtLayer(width, kernel_size=3)

        self.conv3 = nn.Conv2d(width, outplanes, kernel_size=1, bias=False)
        self.bn3 = norm_layer(outplanes)
        self.se = None
        self.act3 = act_layer(inplace=True)
        self.downsample = downsample
        self.stride = stride
        self.dilation = dilation
        self.drop_block = drop_block
        self.drop_path = drop_path

    def zero_init_last_bn(self):
        nn.init.zeros_(self.bn3.weight)

    def forward(self, x):
        if cot_layer_fused.ENABLED and not self.training and cot_layer_fused.eval_block_eligible(self, x):
            return cot_layer_fused.eval_block_forward(self, x)  # inference: the block's launches back to back, nothing kept
        if cot_layer_fused.ENABLED and cot_layer_fused.cm_block_eligible(self, x):
            return cot_layer_fused.cm_block_forward(self, x)  # deep-stage identity block: one node, 1x1 operands channel-major
        if cot_layer_fused.ENABLED and cot_layer_fused.block_eligible(self, x):
            return cot_layer_fused.block_forward(self, x)  # the whole block as one autograd node (opt-in)
        residual = x
        a1, a3 = act_name(self.act1), act_name(self.act3)
        fusable = self.drop_block is None and a1 is not False and a3 is not False
        if fusable:
            x = fused_bn_act(conv1x1(self.conv1, x), self.bn1, a1)
        else:
            x = self.bn1(self.conv1(x))
            if self.drop_block is not None:
                x = self.drop_block(x)
            x = self.act1(x)
        if self.avd is not None:
            x = pool(self.avd, x)
        x = self.conv2(x)
        x = conv1x1(self.conv3, x)
        if fusable and self.drop_path is None:
            if self.downsample is not None:
                residual = run_downsample(self.downsample, residual)
            return fused_bn_act(x, self.bn3, a3, residual)  # bn3 + residual add + act3 in one pass
        x = self.bn3(x)
        if self.drop_block is not None:
            x = self.drop_block(x)
        if self.drop_path is not None:
            x = self.drop_path(x)
        if self.downsample is not None:
            residual = run_downsample(self.downsample, residual)
        x += residual
        return self.act3(x)


def _create_cotnet(variant, pretrained=False, **kwargs):
    return build_model_with_cfg(ResNet, variant, default_cfg=default_cfgs[variant], pretrained=pretrained, **kwargs)


@register_model
def cotnet50(pretrained=False, **kwargs):
    return _create_cotnet("cot_basic", pretrained, block=Bottleneck, layers=[3, 4, 6, 3], **kwargs)


@register_model
def cotnext50_2x48d(pretrained=False, **kwargs):
    return _create_cotnet("cot_basic", pretrained, block=Bottleneck, layers=[3, 4, 6, 3], cardinality=2,
                          base_width=48, **kwargs)


@register_model
def cotnet101(pretrained=False, **kwargs):
    return _create_cotnet("cot_basic", pretrained, block=Bottleneck, layers=[3, 4, 23, 3], **kwargs)


@register_model
def cotnext101_2x48d(pretrained=False, **kwargs):
    return _create_cotnet("cot_basic", pretrained, block=Bottleneck, layers=[3, 4, 23, 3], cardinality=2,
                          base_width=48, **kwargs)

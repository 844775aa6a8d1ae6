"""SE-CoTNetD on MI355X -- drop-in for the reference's models/cotnet_hybrid.py.

`CoTLayer` (ref :48-104) is the same computation as cotnet.CotLayer under a different class name;
`CoTBottleneck` (ref :106-202) uses a SplitAttnConv2d(radix=1) 3x3 conv in the 64/128-wide stages and in the
even-indexed blocks of the 256-wide stage, CoT layers elsewhere (:138,:155); `CoTHybridNet` (ref :331-445) is the
ResNet skeleton without max-pool whose four stages all stride by 2 (:251-256,:431).  Entry points
`se_cotnetd_{50,101,152,152_L,200,270}` (ref :452-498).  Module / parameter names match the reference.
"""
import math

from torch import nn

from . import cot_layer_fused
from .conv1x1 import conv1x1, run_downsample
from .cotnet import CotLayer, _cfg
from .fused_bn import fused_bn_act
from .head_fused import head
from .layers import BlurPool2d, SplitAttnConv2d, create_classifier, get_act_layer
from .registry import build_model_with_cfg, register_model
from .resnet import init_weights, make_blocks, make_stem, stem_forward

default_cfgs = {
    "cot_basic": _cfg(url=""),
    "cot_s": _cfg(url="", input_size=(3, 256, 256), pool_size=(8, 8), crop_pct=0.888, interpolation="bicubic"),
    "cot_m": _cfg(url="", input_size=(3, 288, 288), pool_size=(9, 9), crop_pct=0.9, interpolation="bicubic"),
    "cot_l": _cfg(url="", input_size=(3, 320, 320), pool_size=(10, 10), crop_pct=0.909, interpolation="bicubic"),
}


class CoTLayer(CotLayer):
    """same sub-modules and forward as cotnet.CotLayer (the reference duplicates the class body)"""


class CoTBottleneck(nn.Module):
    expansion = 4

    def __init__(self, block_idx, inplanes, planes, stride=1, downsample=None, cardinality=1, base_width=64,
                 reduce_first=1, dilation=1, first_dilation=None, act_layer=nn.ReLU, norm_layer=nn.BatchNorm2d,
                 attn_layer=None, aa_layer=None, drop_block=None, drop_path=None, radix=1, avd=False, avd_first=True,
                 conv_dim={}, c4_dim=-1, c4_idx={}):
        super(CoTBottleneck, self).__init__()
        width = int(math.floor(planes * (base_width / 64)) * cardinality)
        first_planes = width // reduce_first
        outplanes = planes * self.expansion
        first_dilation = first_dilation or dilation
        self.avd_first = avd_first
        self.avd = None

        self.conv1 = nn.Conv2d(inplanes, first_planes, kernel_size=1, bias=False)
        self.bn1 = norm_layer(first_planes)
        self.act1 = nn.ReLU(inplace=True)

        def pool(s):
            return nn.AvgPool2d(3, s, padding=1) if aa_layer is None else aa_layer(channels=width, stride=s)

        if (width in conv_dim) or (width == c4_dim and block_idx not in c4_idx):
            if stride > 1 and avd:
                self.avd = pool(stride)
                stride = 1
            if radix >= 1:
                self.conv2 = SplitAttnConv2d(first_planes, width, kernel_size=3, stride=stride, padding=first_dilation,
                                             reduction_factor=4, dilation=first_dilation, groups=cardinality,
                                             radix=radix, norm_layer=norm_layer, drop_block=drop_block,
                                             act_layer=get_act_layer("swish"))
            else:
                self.conv2 = nn.Sequential(
                    nn.Conv2d(first_planes, width, kernel_size=3, stride=stride, padding=first_dilation,
                              dilation=first_dilation, groups=cardinality, bias=False),
                    norm_layer(width), act_layer(inplace=True))
        else:
            self.conv2 = CoTLayer(width, kernel_size=3)
            if stride > 1:
                self.avd = pool(stride)

        self.conv3 = nn.Conv2d(width, outplanes, kernel_size=1, bias=False)
        self.bn3 = norm_layer(outplanes)
        self.act3 = nn.ReLU(inplace=True)
        self.downsample = downsample
        self.stride = stride
        self.dilation = dilation
        self.drop_block = drop_block
        self.drop_path = drop_path

    def zero_init_last_bn(self):
        nn.init.zeros_(self.bn3.weight)

    def forward(self, x):
        if cot_layer_fused.ENABLED:  # the whole block as one autograd node (identity-shortcut / AvgPool-first blocks of both kinds)
            if isinstance(self.conv2, CotLayer):
                if cot_layer_fused.block_eligible(self, x):
                    return cot_layer_fused.block_forward(self, x)
            elif cot_layer_fused.sa_block_eligible(self, x):
                return cot_layer_fused.sa_block_forward(self, x)
        residual = x
        if self.drop_block is None:
            x = fused_bn_act(conv1x1(self.conv1, x), self.bn1, "relu")  # act1 is hard-wired ReLU (ref :124)
        else:
            x = self.act1(self.drop_block(self.bn1(self.conv1(x))))
        if self.avd is not None and self.avd_first:
            x = self.avd(x)
        x = self.conv2(x)
        if self.avd is not None and not self.avd_first:
            x = self.avd(x)
        x = conv1x1(self.conv3, x)
        if self.drop_block is None and self.drop_path is None:
            if self.downsample is not None:
                residual = run_downsample(self.downsample, residual)
            return fused_bn_act(x, self.bn3, "relu", residual)  # act3 is hard-wired ReLU (ref :167)
        x = self.bn3(x)
        if self.drop_block is not None:
            x = self.drop_block(x)
        if self.drop_path is not None:
            x = self.drop_path(x)
        if self.downsample is not None:
            residual = run_downsample(self.downsample, residual)
        x += residual
        return self.act3(x)


class CoTHybridNet(nn.Module):
    def __init__(self, block, layers, num_classes=1000, in_chans=3, cardinality=1, base_width=64, stem_width=64,
                 stem_type="", output_stride=32, block_reduce_first=1, down_kernel_size=1, avg_down=False,
                 act_layer=nn.ReLU, norm_layer=nn.BatchNorm2d, aa_layer=None, drop_rate=0.0, drop_path_rate=0.0,
                 drop_block_rate=0.0, global_pool="avg", zero_init_last_bn=True, block_args=None):
        block_args = block_args or dict()
        assert output_stride in (8, 16, 32)
        assert not isinstance(block, list), "per-stage block lists (make_blocks_arr) are unused by the entry points"
        super(CoTHybridNet, self).__init__()
        self.num_classes = num_classes
        self.drop_rate = drop_rate

        # the stem's activations are hard-wired ReLU whatever act_layer says (ref :362-368,:373)
        self.conv1, inplanes = make_stem(in_chans, stem_width, stem_type, norm_layer, nn.ReLU)
        self.bn1 = norm_layer(inplanes)
        self.act1 = nn.ReLU(inplace=True)
        self.feature_info = [dict(num_chs=inplanes, reduction=2, module="act1")]

        stages, finfo = make_blocks(block, [64, 128, 256, 512], layers, inplanes, cardinality=cardinality,
                                    base_width=base_width, output_stride=output_stride,
                                    reduce_first=block_reduce_first, avg_down=avg_down,
                                    down_kernel_size=down_kernel_size, act_layer=act_layer, norm_layer=norm_layer,
                                    aa_layer=aa_layer, drop_block_rate=drop_block_rate,
                                    drop_path_rate=drop_path_rate, first_stride=2, net_stride=2,
                                    pass_block_idx=True, **block_args)
        for name, stage in stages:
            self.add_module(name, stage)
        self.feature_info.extend(finfo)

        self.num_features = 512 * block.expansion
        self.global_pool, self.fc = create_classifier(self.num_features, self.num_classes, pool_type=global_pool)
        init_weights(self, zero_init_last_bn)

    def get_classifier(self):
        return self.fc

    def reset_classifier(self, num_classes, global_pool="avg"):
        self.num_classes = num_classes
        self.global_pool, self.fc = create_classifier(self.num_features, self.num_classes, pool_type=global_pool)

    def forward_features(self, x):
        x = stem_forward(self.conv1, self.bn1, self.act1, x)  # no max-pool (ref :431)
        return self.layer4(self.layer3(self.layer2(self.layer1(x))))

    def forward(self, x):
        x = self.forward_features(x)
        # fc(dropout(global_pool(x))) (models/cotnet_hybrid.py:439-445), on the library's kernels when eligible (head_fused.py)
        return head(self.global_pool, self.fc, x, float(self.drop_rate) if (self.drop_rate and self.training) else 0.0)


def _create_se_cotnetd(variant, pretrained=False, **kwargs):
    return build_model_with_cfg(CoTHybridNet, variant, default_cfg=default_cfgs[variant], pretrained=pretrained,
                                **kwargs)


def _hybrid_args(layers, stem_width, aa_layer, avd, avd_first, c4_blocks):
    return dict(block=CoTBottleneck, layers=layers, act_layer=get_act_layer("swish"), stem_type="deep",
                stem_width=stem_width, avg_down=True, base_width=64, cardinality=1, aa_layer=aa_layer,
                block_args=dict(radix=1, avd=avd, avd_first=avd_first, conv_dim={64, 128}, c4_dim=256,
                                c4_idx=set(range(0, c4_blocks, 2))))


@register_model
def se_cotnetd_50(pretrained=False, **kwargs):
    return _create_se_cotnetd("cot_basic", pretrained, **_hybrid_args([3, 4, 6, 3], 32, None, False, True, 6), **kwargs)


@register_model
def se_cotnetd_101(pretrained=False, **kwargs):
    return _create_se_cotnetd("cot_basic", pretrained, **_hybrid_args([3, 4, 23, 3], 64, None, False, True, 23),
                              **kwargs)


@register_model
def se_cotnetd_152(pretrained=False, **kwargs):
    return _create_se_cotnetd("cot_s", pretrained, **_hybrid_args([3, 8, 36, 3], 64, BlurPool2d, True, False, 36),
                              **kwargs)


@register_model
def se_cotnetd_152_L(pretrained=False, **kwargs):
    return _create_se_cotnetd("cot_l", pretrained, **_hybrid_args([3, 8, 36, 3], 64, BlurPool2d, True, False, 36),
                              **kwargs)


@register_model
def se_cotnetd_200(pretrained=False, **kwargs):
    return _create_se_cotnetd("cot_s", pretrained, **_hybrid_args([3, 24, 36, 3], 64, BlurPool2d, True, False, 36),
                              **kwargs)


@register_model
def se_cotnetd_270(pretrained=False, **kwargs):
    return _create_se_cotnetd("cot_s", pretrained, **_hybrid_args([4, 29, 53, 4], 64, BlurPool2d, True, False, 53),
                              **kwargs)

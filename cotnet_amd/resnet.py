"""ResNet skeleton the CoTNet / CoTNeXt entry points plug their Bottleneck into.

Thin slice of the reference's models/resnet.py: stem (:519-548), `make_blocks` (:404-445),
`downsample_conv/avg` (:364-394), head + init (:570-584), forward (:593-611).  Module names (conv1, bn1,
act1, maxpool, layer1..4, global_pool, fc; blocks as nn.Sequential children "0","1",...) are the
reference's, so state_dict keys are identical.  DropBlock is not provided (drop_block_rate must be 0 --
every CoT recipe leaves it at 0).
"""
import torch.nn.functional as F
from torch import nn

from . import cot_layer_fused
from .fused_bn import fused_bn_act
from .layers import AvgPool2dSame, DropPath, create_classifier
from .pool3x3 import pool
from .head_fused import head
from .stem7x7 import stem_conv


def get_padding(kernel_size, stride, dilation=1):
    return ((stride - 1) + dilation * (kernel_size - 1)) // 2


def downsample_conv(in_channels, out_channels, kernel_size, stride=1, dilation=1, first_dilation=None,
                    norm_layer=None):
    norm_layer = norm_layer or nn.BatchNorm2d
    kernel_size = 1 if stride == 1 and dilation == 1 else kernel_size
    first_dilation = (first_dilation or dilation) if kernel_size > 1 else 1
    p = get_padding(kernel_size, stride, first_dilation)
    return nn.Sequential(
        nn.Conv2d(in_channels, out_channels, kernel_size, stride=stride, padding=p, dilation=first_dilation,
                  bias=False),
        norm_layer(out_channels))


def downsample_avg(in_channels, out_channels, kernel_size, stride=1, dilation=1, first_dilation=None,
                   norm_layer=None):
    norm_layer = norm_layer or nn.BatchNorm2d
    avg_stride = stride if dilation == 1 else 1
    if stride == 1 and dilation == 1:
        pool = nn.Identity()
    else:
        pool_cls = AvgPool2dSame if avg_stride == 1 and dilation > 1 else nn.AvgPool2d
        pool = pool_cls(2, avg_stride, ceil_mode=True, count_include_pad=False)
    return nn.Sequential(pool, nn.Conv2d(in_channels, out_channels, 1, stride=1, padding=0, bias=False),
                         norm_layer(out_channels))


def make_blocks(block_fn, channels, block_repeats, inplanes, reduce_first=1, output_stride=32, down_kernel_size=1,
                avg_down=False, drop_block_rate=0.0, drop_path_rate=0.0, first_stride=1, net_stride=4,
                pass_block_idx=False, **kwargs):
    """`first_stride`/`net_stride`/`pass_block_idx` cover the one place cotnet_hybrid.py's copy of this
    function differs (every stage strides by 2, net_stride starts at 2, block_fn gets block_idx first:
    cotnet_hybrid.py:251-256,:276)."""
    assert not drop_block_rate, "DropBlock is outside the CoT hot-path scope"
    stages = []
    feature_info = []
    total_blocks = sum(block_repeats)
    net_block_idx = 0
    dilation = prev_dilation = 1
    for stage_idx, (planes, num_blocks) in enumerate(zip(channels, block_repeats)):
        stride = first_stride if stage_idx == 0 else 2
        if net_stride >= output_stride:
            dilation *= stride
            stride = 1
        else:
            net_stride *= stride
        downsample = None
        if stride != 1 or inplanes != planes * block_fn.expansion:
            make_ds = downsample_avg if avg_down else downsample_conv
            downsample = make_ds(in_channels=inplanes, out_channels=planes * block_fn.expansion,
                                 kernel_size=down_kernel_size, stride=stride, dilation=dilation,
                                 first_dilation=prev_dilation, norm_layer=kwargs.get("norm_layer"))
        blocks = []
        for block_idx in range(num_blocks):
            dpr = drop_path_rate * net_block_idx / (total_blocks - 1)  # linear stochastic-depth decay
            args = (block_idx,) if pass_block_idx else ()
            blocks.append(block_fn(*args, inplanes, planes, stride if block_idx == 0 else 1,
                                   downsample if block_idx == 0 else None, first_dilation=prev_dilation,
                                   drop_path=DropPath(dpr) if dpr > 0.0 else None, reduce_first=reduce_first,
                                   dilation=dilation, drop_block=None, **kwargs))
            prev_dilation = dilation
            inplanes = planes * block_fn.expansion
            net_block_idx += 1
        stages.append((f"layer{stage_idx + 1}", nn.Sequential(*blocks)))
        feature_info.append(dict(num_chs=inplanes, reduction=net_stride, module=f"layer{stage_idx + 1}"))
    return stages, feature_info


def make_stem(in_chans, stem_width, stem_type, norm_layer, act_layer):
    """-> (conv1 module, inplanes).  '' = 7x7/2 conv; 'deep*' = three 3x3 convs (resnet.py:519-541)."""
    deep = "deep" in stem_type
    inplanes = stem_width * 2 if deep else 64
    if not deep:
        return nn.Conv2d(in_chans, inplanes, kernel_size=7, stride=2, padding=3, bias=False), inplanes
    c1 = c2 = stem_width
    if "tiered" in stem_type:
        c1 = 3 * (stem_width // 4)
        c2 = stem_width if "narrow" in stem_type else 6 * (stem_width // 4)
    conv1 = nn.Sequential(
        nn.Conv2d(in_chans, c1, 3, stride=2, padding=1, bias=False), norm_layer(c1), act_layer(inplace=True),
        nn.Conv2d(c1, c2, 3, stride=1, padding=1, bias=False), norm_layer(c2), act_layer(inplace=True),
        nn.Conv2d(c2, inplanes, 3, stride=1, padding=1, bias=False))
    return conv1, inplanes


def stem_forward(conv1, bn1, act1, x):
    """act1(bn1(conv1(x))) (resnet.py:593-596; cotnet_hybrid.py:431).  Every BatchNorm + ReLU pair -- the two inside a deep
    stem's nn.Sequential and bn1 / act1 behind it -- is one fused pass of the library (MIOpen's BatchNorm took 0.74 - 1 ms per
    call on SE-CoTNetD's 64 x 64..128 x 160 x 160 stem tensors, six calls per step: gpurun_out/r4v_secot_per_shape.csv); the
    7 x 7 convolution of the plain stem goes through `stem_conv`, a deep stem's stride-2 3 x 3 convolution through
    `stem3x3_conv` (csrc/stem3x3.hip) and its two stride-1 ones through `conv3x3` (csrc/conv_lds.hip, groups = 1)."""
    relu = isinstance(act1, nn.ReLU)
    if isinstance(conv1, nn.Sequential):
        mods = list(conv1)
        i = 0
        while i < len(mods):
            if (i + 2 < len(mods) and isinstance(mods[i], nn.Conv2d) and isinstance(mods[i + 1], nn.BatchNorm2d)
                    and isinstance(mods[i + 2], nn.ReLU)):
                x = fused_bn_act(_deep_stem_conv(mods[i], x), mods[i + 1], "relu")  # (the modules' own arithmetic when not eligible)
                i += 3
            else:
                x = _deep_stem_conv(mods[i], x) if isinstance(mods[i], nn.Conv2d) else mods[i](x)
                i += 1
    else:
        x = stem_conv(conv1, x) if relu else conv1(x)
    return fused_bn_act(x, bn1, "relu") if relu else act1(bn1(x))


def _deep_stem_conv(conv, x):
    """one 3 x 3 convolution of a deep stem on the library's kernels; what they do not cover takes the module and is counted among
    the fallbacks (refused under COT_STRICT_DISPATCH=1) while the library's convolutions are switched on"""
    from . import _lib, conv3x3g, stem3x3
    if conv.in_channels == 3 and conv.stride == (2, 2):
        return stem3x3.stem3x3_conv(conv, x)
    if conv3x3g.MODE == "hip" and conv3x3g.eligible(conv, x):
        return conv3x3g.conv3x3(conv, x)
    if conv3x3g.MODE == "hip":
        _lib.fallback("deep_stem_conv", x, f"-> {conv.out_channels}, kernel {tuple(conv.kernel_size)}, stride {tuple(conv.stride)}")
    return conv(x)


def init_weights(model, zero_init_last_bn=True):
    """kaiming-normal(fan_out, relu) on every Conv2d, BN gamma=1 beta=0, then bn3.weight=0 per block
    (resnet.py:575-584).  GroupNorm / Linear keep torch defaults, as in the reference."""
    for m in model.modules():
        if isinstance(m, nn.Conv2d):
            nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
        elif isinstance(m, nn.BatchNorm2d):
            nn.init.constant_(m.weight, 1.0)
            nn.init.constant_(m.bias, 0.0)
    if zero_init_last_bn:
        for m in model.modules():
            if hasattr(m, "zero_init_last_bn"):
                m.zero_init_last_bn()


class ResNet(nn.Module):
    def __init__(self, block, layers, num_classes=1000, in_chans=3, cardinality=1, base_width=64, stem_width=64,
                 stem_type="", output_stride=32, block_reduce_first=1, down_kernel_size=1, avg_down=False,
                 act_layer=nn.ReLU, norm_layer=nn.BatchNorm2d, aa_layer=None, drop_rate=0.0, drop_path_rate=0.0,
                 drop_block_rate=0.0, global_pool="avg", zero_init_last_bn=True, block_args=None):
        block_args = block_args or dict()
        assert output_stride in (8, 16, 32)
        super().__init__()
        self.num_classes = num_classes
        self.drop_rate = drop_rate

        self.conv1, inplanes = make_stem(in_chans, stem_width, stem_type, norm_layer, act_layer)
        self.bn1 = norm_layer(inplanes)
        self.act1 = act_layer(inplace=True)
        self.feature_info = [dict(num_chs=inplanes, reduction=2, module="act1")]
        if aa_layer is not None:
            self.maxpool = nn.Sequential(nn.MaxPool2d(kernel_size=3, stride=1, padding=1),
                                         aa_layer(channels=inplanes, stride=2))
        else:
            self.maxpool = nn.MaxPool2d(kernel_size=3, stride=2, padding=1)

        stages, finfo = make_blocks(block, [64, 128, 256, 512], layers, inplanes, cardinality=cardinality,
                                    base_width=base_width, output_stride=output_stride,
                                    reduce_first=block_reduce_first, avg_down=avg_down,
                                    down_kernel_size=down_kernel_size, act_layer=act_layer, norm_layer=norm_layer,
                                    aa_layer=aa_layer, drop_block_rate=drop_block_rate,
                                    drop_path_rate=drop_path_rate, **block_args)
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
        cot_layer_fused.prepare_drop_path(self, x)  # (single-node blocks: one vectorised stochastic-depth draw per step)
        x = stem_forward(self.conv1, self.bn1, self.act1, x)  # stem BN + ReLU in one pass over 112x112
        x = pool(self.maxpool, x)
        x = self.layer2(self.layer1(x))
        for stage in (self.layer3, self.layer4):  # (which blocks hand a channel-major tensor to their successor: DESIGN 5.8)
            cot_layer_fused.plan_stage_layouts(stage)
        return self.layer4(self.layer3(x))

    def forward(self, x):
        x = self.forward_features(x)
        # fc(dropout(global_pool(x))) (models/resnet.py:605-611), on the library's kernels when eligible
        return head(self.global_pool, self.fc, x, float(self.drop_rate) if (self.drop_rate and self.training) else 0.0)

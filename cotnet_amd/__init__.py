"""cotnet_amd -- MI355X-native implementation of CoTNet's CoT-block hot path.

Public surface mirrors the reference (JDAI-CV/CoTNet):
    cupy_layers.aggregation_zeropad      -> cotnet_amd.aggregation_zeropad  (LocalConvolution, aggregation_zeropad)
    cupy_layers.aggregation_zeropad_mix  -> cotnet_amd.aggregation_zeropad_mix
    models.cotnet                        -> cotnet_amd.cotnet               (CotLayer, CoXtLayer, Bottleneck, cotnet50 ...)
    models.cotnet_hybrid                 -> cotnet_amd.cotnet_hybrid        (CoTLayer, CoTBottleneck, se_cotnetd_* ...)
    models.factory / models.registry     -> cotnet_amd.registry             (create_model, register_model)
Device code lives in cotnet_amd/csrc (HIP, gfx950) behind the C ABI of include/cotnet_amd.h.
"""
from . import _lib  # noqa: F401
# NB: the functions `aggregation_zeropad` / `aggregation_zeropad_mix` are NOT re-exported here: they would shadow the
# sub-modules of the same name.  Import them from the sub-modules, as with the reference's cupy_layers package.
from . import aggregation_zeropad, aggregation_zeropad_mix  # noqa: F401
from .aggregation_zeropad import AggregationZeropad, LocalConvolution  # noqa: F401
from .aggregation_zeropad_mix import AggregationZeropadMix, LocalConvolutionMix  # noqa: F401
from .cotnet import Bottleneck, CotLayer, CoXtLayer, cotnet50, cotnet101, cotnext50_2x48d, cotnext101_2x48d  # noqa: F401
from .cotnet_hybrid import (CoTBottleneck, CoTHybridNet, CoTLayer, se_cotnetd_50, se_cotnetd_101,  # noqa: F401
                            se_cotnetd_152, se_cotnetd_152_L, se_cotnetd_200, se_cotnetd_270)
from .registry import create_model, list_models, load_checkpoint, register_model  # noqa: F401
from .resnet import ResNet  # noqa: F401

__version__ = "0.1.0"

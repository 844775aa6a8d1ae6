"""Minimal model registry + factory (reference: models/registry.py:18-42, models/factory.py:6-64,
models/helpers.py:23-48,:311-357 -- only what the CoT entry points need).

    @register_model            decorates an entry point `fn(pretrained=False, **kwargs) -> nn.Module`
    create_model(name, ...)    same keyword surface as the reference's factory for the CoT models
    load_checkpoint(model, p)  accepts the reference's .pth.tar layout (`state_dict` / `state_dict_ema`,
                               optional `module.` prefix), strict by default
"""
import fnmatch
import os
from collections import OrderedDict
from copy import deepcopy

import torch

_model_entrypoints = {}
_model_to_module = {}


def register_model(fn):
    _model_entrypoints[fn.__name__] = fn
    _model_to_module[fn.__name__] = fn.__module__.split(".")[-1]
    return fn


def is_model(name):
    return name in _model_entrypoints


def model_entrypoint(name):
    return _model_entrypoints[name]


def list_models(filter="", module=""):
    names = [n for n in _model_entrypoints if (not module or _model_to_module[n] == module)]
    if filter:
        names = fnmatch.filter(names, filter)
    return sorted(names)


def build_model_with_cfg(model_cls, variant, pretrained, default_cfg, **kwargs):
    if pretrained:
        # the reference's default_cfgs carry empty urls for every CoT model (cotnet.py:21-34): there is
        # nothing to download; weights come through `checkpoint_path`.
        raise RuntimeError(f"no pretrained url for '{variant}'; pass checkpoint_path= to create_model")
    for unsupported in ("features_only", "pruned", "out_indices"):
        if kwargs.pop(unsupported, None):
            raise NotImplementedError(f"{unsupported} is outside the CoT hot-path scope")
    model = model_cls(**kwargs)
    model.default_cfg = deepcopy(default_cfg)
    return model


def load_state_dict(checkpoint_path, use_ema=False):
    if not (checkpoint_path and os.path.isfile(checkpoint_path)):
        raise FileNotFoundError(f"no checkpoint at '{checkpoint_path}'")
    ckpt = torch.load(checkpoint_path, map_location="cpu", weights_only=False)
    key = "state_dict"
    if isinstance(ckpt, dict):
        if use_ema and "state_dict_ema" in ckpt:
            key = "state_dict_ema"
        sd = ckpt[key] if key in ckpt else ckpt
    else:
        sd = ckpt
    out = OrderedDict()
    for k, v in sd.items():
        out[k[7:] if k.startswith("module.") else k] = v
    return out


def load_checkpoint(model, checkpoint_path, use_ema=False, strict=True):
    return model.load_state_dict(load_state_dict(checkpoint_path, use_ema), strict=strict)


def create_model(model_name, pretrained=False, num_classes=1000, in_chans=3, checkpoint_path="", scriptable=None,
                 exportable=None, no_jit=None, **kwargs):
    for k in ("bn_tf", "bn_momentum", "bn_eps"):  # popped for non-EfficientNets (factory.py:37-40)
        kwargs.pop(k, None)
    dcr = kwargs.pop("drop_connect_rate", None)
    if dcr is not None and kwargs.get("drop_path_rate") is None:
        kwargs["drop_path_rate"] = dcr
    kwargs = {k: v for k, v in kwargs.items() if v is not None}
    if not is_model(model_name):
        raise RuntimeError("Unknown model (%s)" % model_name)
    model = model_entrypoint(model_name)(pretrained=pretrained, num_classes=num_classes, in_chans=in_chans, **kwargs)
    if checkpoint_path:
        load_checkpoint(model, checkpoint_path)
    return model

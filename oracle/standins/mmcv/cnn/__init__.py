"""Stand-in mmcv.cnn (test-only)."""
import copy

import numpy as np
import torch.nn as nn


def build_activation_layer(cfg):
    cfg = copy.copy(cfg)
    typ = cfg.pop('type')
    return {'ReLU': nn.ReLU, 'GELU': nn.GELU, 'Sigmoid': nn.Sigmoid,
            'LeakyReLU': nn.LeakyReLU}[typ](**cfg)


def build_norm_layer(cfg, num_features, postfix=''):
    cfg = copy.copy(cfg)
    typ = cfg.pop('type')
    cfg.pop('requires_grad', None)
    if typ == 'LN':
        cfg.setdefault('eps', 1e-5)
        return 'ln' + str(postfix), nn.LayerNorm(num_features, **cfg)
    if typ == 'GN':
        return 'gn' + str(postfix), nn.GroupNorm(num_channels=num_features, **cfg)
    if typ == 'BN':
        return 'bn' + str(postfix), nn.BatchNorm2d(num_features, **cfg)
    raise KeyError(typ)


def bias_init_with_prob(prior_prob):
    return float(-np.log((1 - prior_prob) / prior_prob))


def normal_init(module, mean=0, std=1, bias=0):
    if hasattr(module, 'weight') and module.weight is not None:
        nn.init.normal_(module.weight, mean, std)
    if hasattr(module, 'bias') and module.bias is not None:
        nn.init.constant_(module.bias, bias)


class ConvModule(nn.Module):
    """conv -> norm -> act; conv bias iff norm_cfg is None (bias='auto')."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0,
                 dilation=1, groups=1, bias='auto', conv_cfg=None, norm_cfg=None,
                 act_cfg=dict(type='ReLU'), inplace=True, **kwargs):
        super().__init__()
        assert conv_cfg is None or conv_cfg.get('type', 'Conv2d') in ('Conv2d', 'Conv')
        if bias == 'auto':
            bias = norm_cfg is None
        self.conv = nn.Conv2d(in_channels, out_channels, kernel_size, stride=stride,
                              padding=padding, dilation=dilation, groups=groups, bias=bias)
        self.with_norm = norm_cfg is not None
        self.with_activation = act_cfg is not None
        if self.with_norm:
            self.norm_name, norm = build_norm_layer(norm_cfg, out_channels)
            self.add_module(self.norm_name, norm)
        if self.with_activation:
            act_cfg = copy.copy(act_cfg)
            if act_cfg['type'] in ('ReLU', 'LeakyReLU'):
                act_cfg.setdefault('inplace', inplace)
            self.activate = build_activation_layer(act_cfg)

    def forward(self, x):
        x = self.conv(x)
        if self.with_norm:
            x = getattr(self, self.norm_name)(x)
        if self.with_activation:
            x = self.activate(x)
        return x


def build_model_from_cfg(cfg, registry, default_args=None):
    """mmcv.cnn.build_model_from_cfg: registry build (plumbing)."""
    from mmcv.registry import build_from_cfg
    return build_from_cfg(cfg, registry, default_args)

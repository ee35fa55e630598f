from mmcv.registry import Registry, build_from_cfg

MODELS = Registry('models')
HEADS = MODELS
LOSSES = Registry('loss')
NECKS = MODELS


def build_head(cfg):
    return build_from_cfg(cfg, HEADS)


def build_loss(cfg):
    return build_from_cfg(cfg, LOSSES)


def build_neck(cfg):
    return build_from_cfg(cfg, NECKS)

from functools import partial

from mmcv.registry import Registry, build_from_cfg

BBOX_ASSIGNERS = Registry('bbox_assigner')


def build_assigner(cfg, **default_args):
    return build_from_cfg(cfg, BBOX_ASSIGNERS, default_args)


def build_sampler(cfg, **default_args):
    from .bbox.builder import BBOX_SAMPLERS
    return build_from_cfg(cfg, BBOX_SAMPLERS, default_args)


def multi_apply(func, *args, **kwargs):
    pfunc = partial(func, **kwargs) if kwargs else func
    return tuple(map(list, zip(*map(pfunc, *args))))


def reduce_mean(tensor):
    return tensor


class AssignResult:
    """mmdet.core.AssignResult (fields only): what `MaskHungarianAssigner.assign` returns."""

    def __init__(self, num_gts, gt_inds, max_overlaps, labels=None):
        self.num_gts = num_gts
        self.gt_inds = gt_inds
        self.max_overlaps = max_overlaps
        self.labels = labels
        self._extra_properties = {}

    def set_extra_property(self, key, value):
        self._extra_properties[key] = value


class BaseAssigner:
    """mmdet.core.BaseAssigner: abstract base, no behaviour."""

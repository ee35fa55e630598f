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


def bbox_overlaps(bboxes1, bboxes2, mode='iou', is_aligned=False, eps=1e-6):
    """mmdet.core.bbox_overlaps (mmdet 2.18, third-party; restated for mode='iou', is_aligned=False): IoU of [x1, y1, x2, y2] boxes."""
    import torch
    assert mode == 'iou' and not is_aligned
    rows, cols = bboxes1.size(-2), bboxes2.size(-2)
    if rows * cols == 0:
        return bboxes1.new(bboxes1.shape[:-2] + (rows, cols))
    area1 = (bboxes1[..., 2] - bboxes1[..., 0]) * (bboxes1[..., 3] - bboxes1[..., 1])
    area2 = (bboxes2[..., 2] - bboxes2[..., 0]) * (bboxes2[..., 3] - bboxes2[..., 1])
    lt = torch.max(bboxes1[..., :, None, :2], bboxes2[..., None, :, :2])
    rb = torch.min(bboxes1[..., :, None, 2:], bboxes2[..., None, :, 2:])
    wh = (rb - lt).clamp(min=0)
    overlap = wh[..., 0] * wh[..., 1]
    union = area1[..., None] + area2[..., None, :] - overlap
    union = torch.max(union, union.new_tensor([eps]))
    return overlap / union

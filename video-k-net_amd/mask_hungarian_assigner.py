"""`MaskHungarianAssigner` — drop-in for the reference's train-time assigner (knet/det/mask_hungarian_assigner.py:117-274):
same ctor kwargs and `assign(...)` signature / result fields.  The [N x G] cost matrix is computed on the GPU
(`vkn_assign_costs_f32`: the mask contractions run on the gather kernel), the linear sum assignment in libvkn's C++
shortest-augmenting-path solver (`vkn_lsap_f32`, the algorithm scipy uses) — scipy is not needed at run time.

Supported: the costs every shipped config uses — `FocalLossCost` (mmdet 2.18 defaults), `DiceCost(pred_act=True)`,
`MaskCost(pred_act=True)`, `topk=1`, no boundary cost.  Anything else raises NotImplementedError.
"""
import numpy as np
import torch

from . import ops


class AssignResult:
    """Fields of mmdet.core.AssignResult that the reference reads (knet/det/mask_pseudo_sampler.py)."""

    def __init__(self, num_gts, gt_inds, max_overlaps, labels=None):
        self.num_gts = num_gts
        self.gt_inds = gt_inds
        self.max_overlaps = max_overlaps
        self.labels = labels
        self.host_pos_inds = None   # sorted numpy array of the matched prediction indices when the assigner knows them on the host


def _cost_cfg(cfg, kind, allowed, defaults):
    cfg = dict(cfg or {})
    typ = cfg.pop('type', None)
    if typ not in allowed:
        raise NotImplementedError(f'{kind}: type {typ!r} is not provided (shipped configs use {allowed})')
    out = dict(defaults)
    for k, v in cfg.items():
        if k not in out:
            raise NotImplementedError(f'{kind}: option {k!r} is not provided')
        out[k] = v
    return out


class MaskHungarianAssigner:

    def __init__(self, cls_cost=dict(type='ClassificationCost', weight=1.), mask_cost=dict(type='SigmoidCost', weight=1.0),
                 dice_cost=dict(), boundary_cost=None, topk=1):
        if boundary_cost is not None:
            raise NotImplementedError('boundary_cost is not provided (no shipped config sets it)')
        if topk != 1:
            raise NotImplementedError('topk > 1 is not provided (no shipped config sets it)')
        self.cls = _cost_cfg(cls_cost, 'cls_cost', ('FocalLossCost',), dict(weight=1.0, alpha=0.25, gamma=2, eps=1e-12))
        self.mask = _cost_cfg(mask_cost, 'mask_cost', ('MaskCost',), dict(weight=1.0, pred_act=False, act_mode='sigmoid'))
        self.dice = _cost_cfg(dice_cost, 'dice_cost', ('DiceCost',), dict(weight=1.0, pred_act=False, act_mode='sigmoid', eps=1e-3))
        for c, name in ((self.mask, 'MaskCost'), (self.dice, 'DiceCost')):
            if not c['pred_act'] or c['act_mode'] != 'sigmoid':
                raise NotImplementedError(f'{name} needs pred_act=True, act_mode="sigmoid" (every shipped config)')
        self.topk = topk

    def cost_matrix(self, bbox_pred, cls_pred, gt_bboxes, gt_labels):
        """[N, G] device tensor: cls_cost + mask_cost + dice_cost (reference :222-241)."""
        use_cls = self.cls['weight'] != 0 and cls_pred is not None
        return ops.assign_costs(bbox_pred, cls_pred if use_cls else None, gt_bboxes, gt_labels,
                                cls_weight=self.cls['weight'] if use_cls else 0.0, dice_weight=self.dice['weight'],
                                mask_weight=self.mask['weight'], focal_alpha=self.cls['alpha'], focal_gamma=self.cls['gamma'],
                                focal_eps=self.cls['eps'], dice_eps=self.dice['eps'])

    def assign(self, bbox_pred, cls_pred, gt_bboxes, gt_labels, img_meta=None, gt_bboxes_ignore=None, eps=1e-7):
        """bbox_pred = mask logits [N,H,W], gt_bboxes = gt masks [G,H,W] (the reference's argument names).
        -> AssignResult(num_gts, gt_inds [N] (0 = background, else 1-based gt index), None, labels [N] (-1 = none))."""
        assert gt_bboxes_ignore is None, 'Only case when gt_bboxes_ignore is None is supported.'
        num_gts, num_bboxes = gt_bboxes.size(0), bbox_pred.size(0)
        gt_inds = bbox_pred.new_full((num_bboxes,), -1, dtype=torch.long)
        labels = bbox_pred.new_full((num_bboxes,), -1, dtype=torch.long)
        if num_gts == 0 or num_bboxes == 0:
            if num_gts == 0:
                gt_inds[:] = 0
            return AssignResult(num_gts, gt_inds, None, labels=labels)
        cost = self.cost_matrix(bbox_pred, cls_pred, gt_bboxes, gt_labels)
        rows, cols = ops.lsap(cost)                       # one D2H copy of [N, G] floats, C++ solver on the host
        rows_host = rows
        rows = torch.from_numpy(rows).to(bbox_pred.device)
        cols = torch.from_numpy(cols).to(bbox_pred.device)
        gt_inds[:] = 0
        gt_inds[rows] = cols + 1
        labels[rows] = gt_labels[cols].to(labels.dtype)
        res = AssignResult(num_gts, gt_inds, None, labels=labels)
        res.host_pos_inds = np.sort(np.asarray(rows_host, dtype=np.int64))   # the LSAP ran on the host: the sampler needs no device nonzero
        return res


try:  # register beside the reference's class when mmdet is importable (same `type` name, force=True)
    from mmdet.core.bbox.builder import BBOX_ASSIGNERS  # type: ignore
    BBOX_ASSIGNERS.register_module(force=True)(MaskHungarianAssigner)
except Exception:  # noqa: BLE001
    pass

"""`MaskPseudoSampler` / `MaskSamplingResult` — knet/det/mask_pseudo_sampler.py:15-191 (same `BBOX_SAMPLERS` name, same fields):
no sampling happens, the one-to-one assignment is only repackaged as (pos, neg) index sets with the matched ground truth."""
import torch


class MaskSamplingResult:
    """knet/det/mask_pseudo_sampler.py:15-170."""

    def __init__(self, pos_inds, neg_inds, masks, gt_masks, assign_result, gt_flags):
        self.pos_inds = pos_inds
        self.neg_inds = neg_inds
        self.pos_masks = masks[pos_inds]
        # `neg_masks` (reference :33) is a copy of every UNMATCHED prediction — ~100 full-resolution masks per image and stage that
        # nothing downstream reads (target building only asks for their number).  Same field, materialised on first access.
        self._all_masks = masks
        self._neg_masks = None
        self.num_neg = int(neg_inds.shape[0])
        self.pos_is_gt = gt_flags[pos_inds]
        self.num_gts = gt_masks.shape[0]
        self.pos_assigned_gt_inds = assign_result.gt_inds[pos_inds] - 1
        if gt_masks.numel() == 0:
            assert self.pos_assigned_gt_inds.numel() == 0
            self.pos_gt_masks = torch.empty_like(gt_masks)
        else:
            self.pos_gt_masks = gt_masks[self.pos_assigned_gt_inds, :]
        self.pos_gt_labels = assign_result.labels[pos_inds] if assign_result.labels is not None else None
        extra = getattr(assign_result, '_extra_properties', {})
        self.pos_gt_pids = extra['pids'][pos_inds] if 'pids' in extra else None

    @property
    def neg_masks(self):
        if self._neg_masks is None:
            self._neg_masks = self._all_masks[self.neg_inds]
        return self._neg_masks

    @property
    def masks(self):
        return torch.cat([self.pos_masks, self.neg_masks])

    @property
    def info(self):
        return dict(pos_inds=self.pos_inds, neg_inds=self.neg_inds, pos_masks=self.pos_masks, neg_masks=self.neg_masks,
                    pos_is_gt=self.pos_is_gt, num_gts=self.num_gts, pos_assigned_gt_inds=self.pos_assigned_gt_inds)


class MaskPseudoSampler:
    """knet/det/mask_pseudo_sampler.py:173-205."""

    def __init__(self, **kwargs):
        pass

    def sample(self, assign_result, masks, gt_masks, **kwargs):
        host_pos = getattr(assign_result, 'host_pos_inds', None)
        dev_pos = getattr(assign_result, 'device_pos_inds', None)
        if dev_pos is not None:
            # device assignment: the matched predictions are already a sorted device tensor of KNOWN length min(N, G); the unmatched
            # ones are the first N - K entries of a stable sort on "is matched" — fixed shapes, no nonzero(), no synchronisation
            n = assign_result.gt_inds.shape[0]
            pos_inds = dev_pos
            neg_inds = torch.sort((assign_result.gt_inds > 0).to(torch.uint8), stable=True)[1][:n - dev_pos.shape[0]]
        elif host_pos is not None:
            # same index sets as the reference's `nonzero(gt_inds > 0 / == 0).unique()` (:197-200), built from the host copy of
            # the assignment: four device -> host synchronisations fewer per image and stage
            import numpy as np
            n = assign_result.gt_inds.shape[0]
            dev = assign_result.gt_inds.device
            pos_inds = torch.from_numpy(host_pos).to(dev)
            neg_inds = torch.from_numpy(np.setdiff1d(np.arange(n, dtype=np.int64), host_pos, assume_unique=True)).to(dev)
        else:
            pos_inds = torch.nonzero(assign_result.gt_inds > 0, as_tuple=False).squeeze(-1).unique()
            neg_inds = torch.nonzero(assign_result.gt_inds == 0, as_tuple=False).squeeze(-1).unique()
        gt_flags = masks.new_zeros(masks.shape[0], dtype=torch.uint8)
        return MaskSamplingResult(pos_inds, neg_inds, masks, gt_masks, assign_result, gt_flags)

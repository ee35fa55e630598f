"""`MaskPseudoSampler` / `MaskSamplingResult` — the registry name and field names of knet/det/mask_pseudo_sampler.py:15-205.
No sampling happens there: the one-to-one assignment is only repackaged as (pos, neg) index sets with the matched ground truth.

Here the result is a VIEW of the assignment: the two things the training loop reads from every image and stage — the matched
predictions' indices and their ground truth (labels, masks) — are built eagerly (four small gathers); everything else the
reference's class materialises at construction (`neg_inds`, `pos_masks`, `neg_masks` — a copy of ~100 full-resolution masks —,
`pos_is_gt`, `masks`) is computed on first access, because nothing downstream of the K-Net heads reads it."""
import torch


class MaskSamplingResult:

    def __init__(self, assign_result, masks, gt_masks, pos_inds, num_neg, neg_inds=None):
        self._assign, self._masks, self._neg_inds = assign_result, masks, neg_inds
        self.pos_inds = pos_inds                                   # sorted indices of the matched predictions
        self.num_pos, self.num_neg = int(pos_inds.shape[0]), int(num_neg)
        self.num_gts = gt_masks.shape[0]
        self.mask_shape, self.mask_dtype, self.device = tuple(masks.shape[1:]), masks.dtype, masks.device
        self.pos_assigned_gt_inds = assign_result.gt_inds[pos_inds] - 1
        if gt_masks.numel() == 0:
            assert self.num_pos == 0
            self.pos_gt_masks = torch.empty_like(gt_masks)
        else:
            self.pos_gt_masks = gt_masks[self.pos_assigned_gt_inds, :]
        self.pos_gt_labels = assign_result.labels[pos_inds] if assign_result.labels is not None else None
        extra = getattr(assign_result, '_extra_properties', {})
        self.pos_gt_pids = extra['pids'][pos_inds] if 'pids' in extra else None

    # ---- fields of the reference's class nobody on this path reads: on demand
    @property
    def neg_inds(self):
        if self._neg_inds is None:
            # the unmatched predictions = the first N - K entries of a stable sort on "is matched": fixed shapes, no nonzero()
            order = torch.sort((self._assign.gt_inds > 0).to(torch.uint8), stable=True)[1]
            self._neg_inds = order[:self.num_neg]
        return self._neg_inds

    @property
    def pos_masks(self):
        return self._masks[self.pos_inds]

    @property
    def neg_masks(self):
        return self._masks[self.neg_inds]

    @property
    def pos_is_gt(self):
        return self._masks.new_zeros(self.num_pos, dtype=torch.uint8)

    @property
    def masks(self):
        return torch.cat([self.pos_masks, self.neg_masks])

    @property
    def info(self):
        return dict(pos_inds=self.pos_inds, neg_inds=self.neg_inds, pos_masks=self.pos_masks, neg_masks=self.neg_masks,
                    pos_is_gt=self.pos_is_gt, num_gts=self.num_gts, pos_assigned_gt_inds=self.pos_assigned_gt_inds)


class MaskPseudoSampler:

    def __init__(self, **kwargs):
        pass

    def sample(self, assign_result, masks, gt_masks, **kwargs):
        n = assign_result.gt_inds.shape[0]
        dev_pos = getattr(assign_result, 'device_pos_inds', None)
        host_pos = getattr(assign_result, 'host_pos_inds', None)
        if dev_pos is not None:       # device assignment: a sorted device tensor of KNOWN length min(N, G) — no synchronisation
            return MaskSamplingResult(assign_result, masks, gt_masks, dev_pos, n - dev_pos.shape[0])
        if host_pos is not None:      # host LSAP: both index sets from its host copy of the assignment
            import numpy as np
            dev = assign_result.gt_inds.device
            neg = torch.from_numpy(np.setdiff1d(np.arange(n, dtype=np.int64), host_pos, assume_unique=True)).to(dev)
            return MaskSamplingResult(assign_result, masks, gt_masks, torch.from_numpy(host_pos).to(dev), neg.shape[0], neg)
        # a foreign AssignResult: the reference's own route (two synchronising nonzero calls, :197-200)
        pos = torch.nonzero(assign_result.gt_inds > 0, as_tuple=False).squeeze(-1).unique()
        neg = torch.nonzero(assign_result.gt_inds == 0, as_tuple=False).squeeze(-1).unique()
        return MaskSamplingResult(assign_result, masks, gt_masks, pos, neg.shape[0], neg)

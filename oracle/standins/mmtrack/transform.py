"""Stand-in for `mmtrack.transform.outs2results` (mmtrack 0.x, third-party, absent offline) — restated: per-class lists; with
`ids`, box rows are [id, x1, y1, x2, y2, score] and detections with id < 0 are dropped; masks are grouped per class in order."""
import numpy as np
import torch


def outs2results(bboxes=None, labels=None, masks=None, ids=None, num_classes=None, **kwargs):
    assert labels is not None and num_classes is not None
    results = dict()
    if ids is not None:
        valid_inds = ids > -1
        ids = ids[valid_inds]
        labels = labels[valid_inds]
    if bboxes is not None:
        if ids is not None:
            bboxes = bboxes[valid_inds]
            if bboxes.shape[0] == 0:
                bbox_results = [np.zeros((0, 6), dtype=np.float32) for _ in range(num_classes)]
            else:
                b, l_, i_ = (t.cpu().numpy() if isinstance(t, torch.Tensor) else t for t in (bboxes, labels, ids))
                bbox_results = [np.concatenate((i_[l_ == c, None], b[l_ == c, :]), axis=1) for c in range(num_classes)]
        else:
            b, l_ = (t.cpu().numpy() if isinstance(t, torch.Tensor) else t for t in (bboxes, labels))
            bbox_results = [b[l_ == c, :] for c in range(num_classes)]
        results['bbox_results'] = bbox_results
    if masks is not None:
        if ids is not None:
            masks = masks[valid_inds]
        masks = masks.cpu().numpy() if isinstance(masks, torch.Tensor) else masks
        lab = labels.cpu().numpy() if isinstance(labels, torch.Tensor) else labels
        mask_results = [[] for _ in range(num_classes)]
        for i in range(masks.shape[0]):
            mask_results[lab[i]].append(masks[i])
        results['mask_results'] = mask_results
    return results

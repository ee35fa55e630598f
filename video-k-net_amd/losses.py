"""Training losses of the head — torch autograd (they touch [num_pos, H, W] tensors once per stage and are not on the roofline;
SURVEY.md §8 keeps them outside the three HIP ops).

* `FocalLoss`  — mmdet 2.18 `FocalLoss(use_sigmoid=True)` (third-party, restated; mmdet/models/losses/focal_loss.py:
  py_sigmoid_focal_loss + weight_reduce_loss), the classification loss of every shipped config
  (configs/det/_base_/models/knet_kitti_step_s3_r50_fpn.py:131-136).
* `CrossEntropyLoss` — the reference's OWN class, knet/cross_entropy_loss.py:140-221 (registered with force=True over mmdet's):
  `use_sigmoid=True` -> binary_cross_entropy (:61-101), `use_mask` -> mask_cross_entropy (:104-137), else cross_entropy (:8-43).
* `DiceLoss` — mmdet 2.18 `DiceLoss(use_sigmoid=True, activate=True, eps=1e-3)` = the formula the reference keeps (commented out)
  in knet/det/dice_loss.py:8-18: 1 - 2 a / (b + c).
* `accuracy`, `reduce_mean` — mmdet helpers used by `KernelUpdateHead.loss` (knet/det/kernel_update_head.py:297, 310).
"""
import torch
import torch.distributed as dist
import torch.nn as nn
import torch.nn.functional as F


def reduce_loss(loss, reduction):
    if reduction == 'none':
        return loss
    return loss.mean() if reduction == 'mean' else loss.sum()


def weight_reduce_loss(loss, weight=None, reduction='mean', avg_factor=None):
    """mmdet/models/losses/utils.py: element-wise weight, then mean / sum, or sum / avg_factor."""
    if weight is not None:
        loss = loss * weight
    if avg_factor is None:
        return reduce_loss(loss, reduction)
    if reduction == 'mean':
        return loss.sum() / avg_factor
    if reduction != 'none':
        raise ValueError('avg_factor can not be used with reduction="sum"')
    return loss


def reduce_mean(tensor):
    """mmdet.core.reduce_mean: average a tensor over the ranks (identity without an initialised process group)."""
    if not (dist.is_available() and dist.is_initialized()):
        return tensor
    tensor = tensor.clone()
    dist.all_reduce(tensor.div_(dist.get_world_size()), op=dist.ReduceOp.SUM)
    return tensor


def accuracy(pred, target, topk=1, thresh=None):
    """mmdet.models.losses.accuracy for topk=1: percentage of rows whose arg-max equals the target, as a [1] tensor."""
    if pred.size(0) == 0:
        return pred.new_tensor([0.])
    _, label = pred.topk(1, dim=1)
    correct = label.t().eq(target.view(1, -1).expand_as(label.t()))
    if thresh is not None:
        correct = correct & (pred.gather(1, label) > thresh).t()
    return correct[:1].reshape(-1).float().sum(0, keepdim=True).mul_(100.0 / pred.size(0))


def py_sigmoid_focal_loss(pred, target, weight=None, gamma=2.0, alpha=0.25, reduction='mean', avg_factor=None):
    pred_sigmoid = pred.sigmoid()
    target = target.type_as(pred)
    pt = (1 - pred_sigmoid) * target + pred_sigmoid * (1 - target)
    focal_weight = (alpha * target + (1 - alpha) * (1 - target)) * pt.pow(gamma)
    loss = F.binary_cross_entropy_with_logits(pred, target, reduction='none') * focal_weight
    if weight is not None:
        if weight.shape != loss.shape:
            if weight.size(0) == loss.size(0):
                weight = weight.view(-1, 1)
            else:
                assert weight.numel() == loss.numel()
                weight = weight.view(loss.size(0), -1)
        assert weight.ndim == loss.ndim
    return weight_reduce_loss(loss, weight, reduction, avg_factor)


class FocalLoss(nn.Module):
    fused = True      # CUDA tensors with reduction 'mean' and an avg_factor: the fused HIP kernel (False: the torch formula, A/B)

    def __init__(self, use_sigmoid=True, gamma=2.0, alpha=0.25, reduction='mean', loss_weight=1.0):
        super().__init__()
        assert use_sigmoid is True, 'Only sigmoid focal loss supported now.'
        self.use_sigmoid = use_sigmoid
        self.gamma, self.alpha, self.reduction, self.loss_weight = gamma, alpha, reduction, loss_weight

    def forward(self, pred, target, weight=None, avg_factor=None, reduction_override=None):
        assert reduction_override in (None, 'none', 'mean', 'sum')
        reduction = reduction_override if reduction_override else self.reduction
        num_classes = pred.size(1)
        if (self.fused and pred.is_cuda and pred.dtype == torch.float32 and reduction == 'mean' and avg_factor is not None
                and pred.dim() == 2 and target.dim() == 1 and (weight is None or weight.numel() in (pred.size(0), pred.numel()))):
            # one HIP pass for the element losses and their derivative (csrc/vkn_loss.hip); same values (tests/test_gpu_train.py)
            from . import autograd as vag
            return vag.focal_loss(pred, target, weight, self.loss_weight, avg_factor, self.alpha, self.gamma)
        target = F.one_hot(target, num_classes=num_classes + 1)[:, :num_classes]   # label == num_classes is background
        return self.loss_weight * py_sigmoid_focal_loss(pred, target, weight, gamma=self.gamma, alpha=self.alpha,
                                                        reduction=reduction, avg_factor=avg_factor)


def cross_entropy(pred, label, weight=None, reduction='mean', avg_factor=None, class_weight=None, ignore_index=-100):
    loss = F.cross_entropy(pred, label, weight=class_weight, reduction='none', ignore_index=ignore_index)
    if weight is not None:
        weight = weight.float()
    return weight_reduce_loss(loss, weight=weight, reduction=reduction, avg_factor=avg_factor)


def _expand_onehot_labels(labels, label_weights, label_channels):
    bin_labels = labels.new_full((labels.size(0), label_channels), 0)
    inds = torch.nonzero((labels >= 0) & (labels < label_channels), as_tuple=False).squeeze()
    if inds.numel() > 0:
        bin_labels[inds, labels[inds]] = 1
    if label_weights is None:
        return bin_labels, None
    return bin_labels, label_weights.view(-1, 1).expand(label_weights.size(0), label_channels)


def binary_cross_entropy(pred, label, weight=None, reduction='mean', avg_factor=None, class_weight=None):
    if pred.dim() != label.dim():
        label, weight = _expand_onehot_labels(label, weight, pred.size(-1))
    if weight is not None:
        weight = weight.float()
    loss = F.binary_cross_entropy_with_logits(pred, label.float(), pos_weight=class_weight, reduction='none')
    return weight_reduce_loss(loss, weight, reduction=reduction, avg_factor=avg_factor)


def mask_cross_entropy(pred, target, label, reduction='mean', avg_factor=None, class_weight=None):
    assert reduction == 'mean' and avg_factor is None
    num_rois = pred.size()[0]
    inds = torch.arange(0, num_rois, dtype=torch.long, device=pred.device)
    pred_slice = pred[inds, label].squeeze(1)
    return F.binary_cross_entropy_with_logits(pred_slice, target, weight=class_weight, reduction='mean')[None]


class CrossEntropyLoss(nn.Module):

    def __init__(self, use_sigmoid=False, use_mask=False, reduction='mean', class_weight=None, loss_weight=1.0):
        super().__init__()
        assert (use_sigmoid is False) or (use_mask is False)
        self.use_sigmoid, self.use_mask = use_sigmoid, use_mask
        self.reduction, self.loss_weight, self.class_weight = reduction, loss_weight, class_weight
        self.cls_criterion = binary_cross_entropy if use_sigmoid else (mask_cross_entropy if use_mask else cross_entropy)

    def forward(self, cls_score, label, weight=None, avg_factor=None, reduction_override=None, **kwargs):
        assert reduction_override in (None, 'none', 'mean', 'sum')
        reduction = reduction_override if reduction_override else self.reduction
        class_weight = cls_score.new_tensor(self.class_weight) if self.class_weight is not None else None
        return self.loss_weight * self.cls_criterion(cls_score, label, weight, class_weight=class_weight,
                                                     reduction=reduction, avg_factor=avg_factor, **kwargs)


def dice_loss(pred, target, weight=None, eps=1e-3, reduction='mean', avg_factor=None):
    inp = pred.flatten(1)
    target = target.flatten(1).float()
    a = torch.sum(inp * target, 1)
    b = torch.sum(inp * inp, 1) + eps
    c = torch.sum(target * target, 1) + eps
    loss = 1 - (2 * a) / (b + c)
    if weight is not None:
        assert weight.ndim == loss.ndim and len(weight) == len(pred)
    return weight_reduce_loss(loss, weight, reduction, avg_factor)


class DiceLoss(nn.Module):

    def __init__(self, use_sigmoid=True, activate=True, reduction='mean', loss_weight=1.0, eps=1e-3):
        super().__init__()
        self.use_sigmoid, self.activate = use_sigmoid, activate
        self.reduction, self.loss_weight, self.eps = reduction, loss_weight, eps

    def forward(self, pred, target, weight=None, reduction_override=None, avg_factor=None):
        assert reduction_override in (None, 'none', 'mean', 'sum')
        reduction = reduction_override if reduction_override else self.reduction
        if self.activate:
            if not self.use_sigmoid:
                raise NotImplementedError
            pred = pred.sigmoid()
        return self.loss_weight * dice_loss(pred, target, weight, eps=self.eps, reduction=reduction, avg_factor=avg_factor)

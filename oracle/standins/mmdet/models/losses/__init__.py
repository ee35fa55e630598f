"""mmdet 2.18 losses the reference's TRAINING path builds through `build_loss` — third-party arithmetic that is absent offline,
restated from the published mmdet 2.18 sources (mmdet/models/losses/{focal_loss,dice_loss,accuracy,utils}.py):
  * FocalLoss(use_sigmoid=True): py_sigmoid_focal_loss (the CPU path; the CUDA op computes the same function) + weight_reduce_loss
  * DiceLoss(use_sigmoid=True, activate=True, eps=1e-3): 1 - 2 a / (b + c) per row, weight_reduce_loss
  * accuracy(pred, target): top-1 percentage as a [1] tensor
`CrossEntropyLoss` is NOT restated: the reference ships its own (knet/cross_entropy_loss.py, registered with force=True), which
oracle/gen_golden.py imports unmodified; the class below is only the mmdet default it overrides."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from mmdet.models.builder import LOSSES
from mmdet.models.losses.utils import weight_reduce_loss


def accuracy(pred, target, topk=1, thresh=None):
    assert topk == 1
    if pred.size(0) == 0:
        return pred.new_tensor([0.])
    _, pred_label = pred.topk(1, dim=1)
    pred_label = pred_label.t()
    correct = pred_label.eq(target.view(1, -1).expand_as(pred_label))
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


@LOSSES.register_module()
class FocalLoss(nn.Module):
    def __init__(self, use_sigmoid=True, gamma=2.0, alpha=0.25, reduction='mean', loss_weight=1.0):
        super().__init__()
        assert use_sigmoid is True
        self.use_sigmoid, self.gamma, self.alpha, self.reduction, self.loss_weight = use_sigmoid, gamma, alpha, reduction, loss_weight

    def forward(self, pred, target, weight=None, avg_factor=None, reduction_override=None):
        reduction = reduction_override if reduction_override else self.reduction
        num_classes = pred.size(1)
        target = F.one_hot(target, num_classes=num_classes + 1)[:, :num_classes]
        return self.loss_weight * py_sigmoid_focal_loss(pred, target, weight, gamma=self.gamma, alpha=self.alpha,
                                                        reduction=reduction, avg_factor=avg_factor)


def dice_loss(pred, target, weight=None, eps=1e-3, reduction='mean', avg_factor=None):
    inp = pred.flatten(1)
    target = target.flatten(1).float()
    a = torch.sum(inp * target, 1)
    b = torch.sum(inp * inp, 1) + eps
    c = torch.sum(target * target, 1) + eps
    loss = 1 - (2 * a) / (b + c)
    return weight_reduce_loss(loss, weight, reduction, avg_factor)


@LOSSES.register_module()
class DiceLoss(nn.Module):
    def __init__(self, use_sigmoid=True, activate=True, reduction='mean', loss_weight=1.0, eps=1e-3):
        super().__init__()
        self.use_sigmoid, self.activate, self.reduction, self.loss_weight, self.eps = use_sigmoid, activate, reduction, loss_weight, eps

    def forward(self, pred, target, weight=None, reduction_override=None, avg_factor=None):
        reduction = reduction_override if reduction_override else self.reduction
        if self.activate:
            assert self.use_sigmoid
            pred = pred.sigmoid()
        return self.loss_weight * dice_loss(pred, target, weight, eps=self.eps, reduction=reduction, avg_factor=avg_factor)


@LOSSES.register_module()
class CrossEntropyLoss(nn.Module):
    """Shell of mmdet's default; replaced (force=True) by knet/cross_entropy_loss.py when gen_golden.py imports it."""

    def __init__(self, use_sigmoid=False, **kwargs):
        super().__init__()
        self.use_sigmoid = use_sigmoid

    def forward(self, *a, **k):
        raise NotImplementedError('stand-in shell: import knet.cross_entropy_loss for the reference\'s own CrossEntropyLoss')

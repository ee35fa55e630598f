"""Loss *shells*: the hot path only reads `.use_sigmoid`; no loss arithmetic is restated."""
import torch.nn as nn

from mmdet.models.builder import LOSSES


def accuracy(pred, target, topk=1, thresh=None):
    raise NotImplementedError('stand-in: training metrics are out of scope')


class _Shell(nn.Module):
    def __init__(self, use_sigmoid=False, **kwargs):
        super().__init__()
        self.use_sigmoid = use_sigmoid

    def forward(self, *a, **k):
        raise NotImplementedError('stand-in: loss arithmetic is out of scope')


for _n in ('FocalLoss', 'CrossEntropyLoss', 'DiceLoss'):
    LOSSES.register_module(name=_n, module=type(_n, (_Shell,), {}))

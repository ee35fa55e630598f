from abc import ABCMeta

import torch.nn as nn


class BaseRoIHead(nn.Module, metaclass=ABCMeta):
    """Ctor contract of mmdet 2.18 BaseRoIHead: init_bbox_head / init_mask_head / init_assigner_sampler."""

    def __init__(self, bbox_roi_extractor=None, bbox_head=None, mask_roi_extractor=None,
                 mask_head=None, shared_head=None, train_cfg=None, test_cfg=None,
                 pretrained=None, init_cfg=None):
        super().__init__()
        self.train_cfg = train_cfg
        self.test_cfg = test_cfg
        if bbox_head is not None:
            self.init_bbox_head(bbox_roi_extractor, bbox_head)
        if mask_head is not None:
            self.init_mask_head(mask_roi_extractor, mask_head)
        self.init_assigner_sampler()

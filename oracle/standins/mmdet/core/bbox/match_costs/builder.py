"""Stand-in mmdet.core.bbox.match_costs.builder (test-only): the MATCH_COST registry and mmdet 2.18's FocalLossCost, restated
(third-party code that is absent offline; the K-Net configs use it with its defaults, weight=2.0 —
configs/det/_base_/models/knet_kitti_step_s3_r50_fpn.py:145)."""
from mmcv.registry import Registry, build_from_cfg

MATCH_COST = Registry('match_cost')


def build_match_cost(cfg, default_args=None):
    return build_from_cfg(cfg, MATCH_COST, default_args)


@MATCH_COST.register_module()
class FocalLossCost:
    """cost[n][g] = weight * (pos - neg)[n][gt_labels[g]] with p = sigmoid(cls_pred),
    neg = -log(1 - p + eps) (1 - alpha) p^gamma,  pos = -log(p + eps) alpha (1 - p)^gamma."""

    def __init__(self, weight=1., alpha=0.25, gamma=2, eps=1e-12):
        self.weight = weight
        self.alpha = alpha
        self.gamma = gamma
        self.eps = eps

    def __call__(self, cls_pred, gt_labels):
        cls_pred = cls_pred.sigmoid()
        neg_cost = -(1 - cls_pred + self.eps).log() * (1 - self.alpha) * cls_pred.pow(self.gamma)
        pos_cost = -(cls_pred + self.eps).log() * self.alpha * (1 - cls_pred).pow(self.gamma)
        cls_cost = pos_cost[:, gt_labels] - neg_cost[:, gt_labels]
        return cls_cost * self.weight

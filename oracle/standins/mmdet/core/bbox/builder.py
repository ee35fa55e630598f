from mmcv.registry import Registry

BBOX_SAMPLERS = Registry('bbox_sampler')

from mmdet.core import BBOX_ASSIGNERS  # noqa: E402,F401  (one registry object, as in mmdet)

from mmcv.registry import Registry

BBOX_SAMPLERS = Registry('bbox_sampler')

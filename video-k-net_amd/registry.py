"""Registry surface of the reference.

The reference registers `KernelIterHead` / `KernelUpdateHead` / `VideoKernel*Head` in mmdet's `HEADS` (= `MODELS` in mmdet 2.18;
knet/det/kernel_iter_head.py:5,11, knet/det/kernel_update_head.py:9,16, knet/video/*.py) and `KernelUpdator` in mmcv's
`TRANSFORMER_LAYER` (knet/kernel_updator.py:4,7), and builds them from config dicts with `build_head` /
`build_transformer_layer` / `build_loss`.  When mmdet/mmcv are importable our classes register THERE (force=True, so that
`custom_imports=['video_k_net_amd']` replaces the reference classes under the unchanged configs); otherwise a bundled
registry with the same `register_module()/build(cfg)` behaviour is used.
"""
import copy

import torch.nn as nn


class Registry:
    """mmcv.utils.Registry look-alike: `@R.register_module()` and `R.build(cfg)` with `type=` dispatch."""

    def __init__(self, name):
        self.name = name
        self._module_dict = {}

    def get(self, key):
        return self._module_dict.get(key)

    def __contains__(self, key):
        return key in self._module_dict

    def register_module(self, name=None, force=False, module=None):
        def _reg(cls):
            key = name or cls.__name__
            if key in self._module_dict and not force:
                raise KeyError(f'{key} is already registered in {self.name}')
            self._module_dict[key] = cls
            return cls
        return _reg(module) if module is not None else _reg

    def build(self, cfg, default_args=None):
        if not isinstance(cfg, dict) or 'type' not in cfg:
            raise TypeError(f'cfg must be a dict with a "type" key, got {cfg!r}')
        args = copy.copy(dict(cfg))
        for k, v in (default_args or {}).items():
            args.setdefault(k, v)
        typ = args.pop('type')
        cls = self.get(typ) if isinstance(typ, str) else typ
        if cls is None:
            raise KeyError(f'{typ} is not in the {self.name} registry')
        return cls(**args)


try:  # real mmdet / mmcv present: be a plug-in of the reference's own registries
    from mmcv.cnn.bricks.transformer import TRANSFORMER_LAYER, build_transformer_layer  # type: ignore
    from mmdet.models.builder import HEADS, build_head, build_loss  # type: ignore
    from mmdet.models.roi_heads import BaseRoIHead  # type: ignore
    from mmdet.core import build_assigner, build_sampler  # type: ignore
    HAVE_MM = True
except Exception:  # noqa: BLE001  (mmcv/mmdet are not installed in the build image)
    HAVE_MM = False
    HEADS = Registry('models')
    TRANSFORMER_LAYER = Registry('transformerLayer')
    LOSSES = Registry('loss')
    BBOX_ASSIGNERS = Registry('bbox_assigner')
    BBOX_SAMPLERS = Registry('bbox_sampler')

    def build_assigner(cfg, **default_args):
        return BBOX_ASSIGNERS.build(cfg, default_args)

    def build_sampler(cfg, **default_args):
        default_args.pop('context', None)      # mmdet passes the head as `context`; the pseudo sampler ignores it
        return BBOX_SAMPLERS.build(cfg, default_args)

    def build_head(cfg):
        return HEADS.build(cfg)

    def build_transformer_layer(cfg, default_args=None):
        return TRANSFORMER_LAYER.build(cfg, default_args)

    def build_loss(cfg):
        return LOSSES.build(cfg)

    class BaseRoIHead(nn.Module):
        """Ctor contract of mmdet 2.18 `BaseRoIHead` (init_bbox_head / init_mask_head / init_assigner_sampler order,
        knet/det/kernel_iter_head.py:67-68,85-116)."""

        def __init__(self, bbox_roi_extractor=None, bbox_head=None, mask_roi_extractor=None, mask_head=None,
                     shared_head=None, train_cfg=None, test_cfg=None, pretrained=None, init_cfg=None):
            super().__init__()
            self.train_cfg = train_cfg
            self.test_cfg = test_cfg
            if bbox_head is not None:
                self.init_bbox_head(bbox_roi_extractor, bbox_head)
            if mask_head is not None:
                self.init_mask_head(mask_roi_extractor, mask_head)
            self.init_assigner_sampler()


def _register_training_components():
    """Losses / assigner / sampler of the training path.  Bundled registries: all of ours.  With mmdet present: OUR assigners and
    sampler replace the reference's in mmdet's own `BBOX_ASSIGNERS` / `BBOX_SAMPLERS` (force=True — our heads call
    `assigner.assign_batch` and read device-side results, an interface the reference's `MaskHungarianAssigner`
    (knet/det/mask_hungarian_assigner.py:188-274, CPU scipy) does not have); the reference's own CrossEntropyLoss override is
    mirrored (force=True); FocalLoss / DiceLoss stay mmdet's (registered here only where mmdet lacks the name)."""
    from . import losses
    from .mask_hungarian_assigner import MaskHungarianAssigner, MaskHungarianAssignerVideo
    from .mask_pseudo_sampler import MaskPseudoSampler
    if HAVE_MM:
        from mmdet.core.bbox.builder import BBOX_ASSIGNERS as mm_assigners, BBOX_SAMPLERS as mm_samplers  # type: ignore
        from mmdet.models.builder import LOSSES as mm_losses  # type: ignore
        try:
            import mmdet.models.losses  # noqa: F401  (mmdet's own losses register on import; `mmdet.models` does this itself in 2.18)
        except ImportError:
            pass
        for cls in (MaskHungarianAssigner, MaskHungarianAssignerVideo):
            mm_assigners.register_module(force=True)(cls)
        mm_samplers.register_module(force=True)(MaskPseudoSampler)
        # CrossEntropyLoss: the reference overrides mmdet's with its own (knet/cross_entropy_loss.py:139, `register_module(force=True)`)
        # — so does this package, with its restatement of THAT class; FocalLoss / DiceLoss are mmdet's own in the reference
        mm_losses.register_module(force=True)(losses.CrossEntropyLoss)
        for cls in (losses.FocalLoss, losses.DiceLoss):
            if mm_losses.get(cls.__name__) is None:
                mm_losses.register_module()(cls)
        return
    for cls in (losses.FocalLoss, losses.CrossEntropyLoss, losses.DiceLoss):
        LOSSES.register_module(force=True)(cls)
    BBOX_ASSIGNERS.register_module(force=True)(MaskHungarianAssigner)
    BBOX_ASSIGNERS.register_module(force=True)(MaskHungarianAssignerVideo)
    BBOX_SAMPLERS.register_module(force=True)(MaskPseudoSampler)


def register_head(cls):
    return HEADS.register_module(force=True)(cls)


def register_transformer_layer(cls):
    return TRANSFORMER_LAYER.register_module(force=True)(cls)

"""video_k_net_amd — MI355X-native kernel-update head for Video K-Net (directory `video-k-net_amd/`).

Importing this package registers `KernelUpdator` (TRANSFORMER_LAYER) and `KernelUpdateHead`, `VideoKernelUpdateHead`,
`KernelIterHead`, `VideoKernelIterHead`, `ConvKernelHead` (HEADS) — into mmcv/mmdet's registries when they are importable, else into the bundled
ones — so the reference's config dicts build these classes unchanged.  All arithmetic runs in
`lib/libvkn.so` (hand-written HIP for gfx950, C ABI in include/vkn.h); there is no CPU fallback.
"""
from . import _lib, configs, ops, registry  # noqa: F401
from ._lib import VknError, VknLibraryError, build  # noqa: F401
from .kernel_updator import KernelUpdator  # noqa: F401
from .kernel_update_head import KernelUpdateHead, VideoKernelUpdateHead  # noqa: F401
from .kernel_iter_head import KernelIterHead, VideoKernelIterHead  # noqa: F401
from .kernel_head import ConvKernelHead, ConvKernelHeadVideo  # noqa: F401
from .knet_vis import KernelFrameIterHeadVideo, KernelIterHeadVideo, KernelUpdateHeadVideo  # noqa: F401
from .mask_hungarian_assigner import MaskHungarianAssigner, MaskHungarianAssignerVideo  # noqa: F401
from .qd_tracker import QuasiDenseEmbedTracker, build_tracker  # noqa: F401
from .track_heads import QuasiDenseMaskEmbedHeadGTMask  # noqa: F401
from .registry import HEADS, TRANSFORMER_LAYER, build_head, build_transformer_layer  # noqa: F401
from . import autograd, losses  # noqa: F401
from .mask_pseudo_sampler import MaskPseudoSampler  # noqa: F401

registry._register_training_components()

__all__ = ['KernelUpdator', 'KernelUpdateHead', 'VideoKernelUpdateHead', 'KernelIterHead', 'VideoKernelIterHead', 'ConvKernelHead', 'MaskHungarianAssigner', 'MaskHungarianAssignerVideo',
           'HEADS', 'TRANSFORMER_LAYER', 'build_head', 'build_transformer_layer', 'ops', 'build', 'VknError',
           'VknLibraryError']

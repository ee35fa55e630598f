"""Stand-in mmcv.utils (test-only): Registry."""
from mmcv.registry import Registry, build_from_cfg  # noqa: F401

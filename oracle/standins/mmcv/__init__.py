"""Stand-in mmcv (test-only). See oracle/standins/README.md."""
from .registry import Registry, build_from_cfg  # noqa: F401

"""Stand-in mmdet.core.bbox.match_costs (test-only)."""
from .builder import MATCH_COST, FocalLossCost, build_match_cost  # noqa: F401

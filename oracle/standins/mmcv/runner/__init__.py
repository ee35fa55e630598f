"""Stand-in mmcv.runner (test-only)."""


def force_fp32(apply_to=None, out_fp16=False):
    def deco(fn):
        return fn
    return deco

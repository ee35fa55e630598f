"""Stand-in for `unitrack.mask` (the real module pulls cv2 / pycocotools / torchvision, none installable here).
`tensor_mask2box` restates /root/reference/unitrack/utils/mask.py:40-45,80-90 (coords2bbox_all over `mask.nonzero()`): per mask
(xmin, ymin, xmax, ymax) of its non-zero pixels, (-1, -1, 10, 10) for an empty mask."""
import numpy as np


def mask2box(*a, **k):
    raise NotImplementedError('stand-in: result formatting is out of scope')


def tensor_mask2box(masks):
    boxes = []
    for mask in masks:
        m = mask.nonzero().float()
        if m.numel() > 0:
            box = (m[:, 1].min().item(), m[:, 0].min().item(), m[:, 1].max().item(), m[:, 0].max().item())
        else:
            box = (-1, -1, 10, 10)
        boxes.append(box)
    return np.asarray(boxes)

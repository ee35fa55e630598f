#!/usr/bin/env python3
"""GPU diagnostic: time of the fused post-head pipeline (vkn_panoptic_joint_f32) at BASELINE cfg2 geometry, next to the plain
torch evaluation of the same chain (which materialises K x Ho x Wo fp32 per frame)."""
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import vkn_import  # noqa: E402
from oracle import synth  # noqa: E402  (diagnostic input generator only)

vkn = vkn_import.load()
dev = 'cuda:0'
B, N, Np, T, ncls, Hm, Wm, up = int(os.environ.get('B', 8)), 117, 100, 2, 19, 128, 256, 4
cls_np, logit_np = synth.panoptic_inputs(B, N, Np, ncls, Hm, Wm, 5)
cls, logits = torch.from_numpy(cls_np).to(dev), torch.from_numpy(logit_np).to(dev)
shape = (1024, 2048)


def timeit(fn, reps=10, warm=2):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


ms = timeit(lambda: vkn.ops.panoptic_joint(cls, logits, Np, T, Np, 0.25, 0.6, shape, shape, shape, upsample_stride=up))
print(f'fused panoptic_joint, B={B}, 1024x2048, K=117: {ms:.3f} ms  ({B / ms * 1e3:.0f} frames/s)')


def torch_chain():
    out = []
    for b in range(B):
        scaled = F.interpolate(logits[b][None], scale_factor=up, mode='bilinear', align_corners=False)[0]
        tm = F.interpolate(scaled[None].sigmoid(), size=shape, mode='bilinear', align_corners=False)[0]
        ids = (cls[b, :, 0].view(-1, 1, 1) * tm).argmax(0)
        out.append((ids, (tm >= 0.5).flatten(1).sum(1), torch.bincount(ids.flatten(), minlength=N)))
    return out


ms_t = timeit(torch_chain, reps=3, warm=1)
print(f'torch ops (no selection / merge loop), same chain: {ms_t:.3f} ms  ({B / ms_t * 1e3:.0f} frames/s)')

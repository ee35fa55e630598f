#!/usr/bin/env python3
"""GPU diagnostic: train-time assignment at the cfg2 assign resolution (256x512), libvkn vs the same cost matrix in torch ops."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import vkn_import  # noqa: E402
from oracle import synth  # noqa: E402  (diagnostic input generator only)

vkn = vkn_import.load()
dev = 'cuda:0'
N, G, ncls, H, W = 100, 40, 2, 256, 512
lo, cl, gt, lab = (torch.from_numpy(a).to(dev) for a in synth.assign_inputs(N, G, ncls, H, W, 3))
a = vkn.MaskHungarianAssigner(cls_cost=dict(type='FocalLossCost', weight=2.0), dice_cost=dict(type='DiceCost', weight=4.0, pred_act=True),
                              mask_cost=dict(type='MaskCost', weight=1.0, pred_act=True))


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


def torch_cost():
    p = lo.sigmoid()
    p1, p2, g = p.clamp(0.001, 1.0).flatten(1), p.clamp(0.01, 1.0), gt
    dice = -(2 * torch.einsum('nh,mh->nm', p1, g.flatten(1))) / ((p1 * p1).sum(1)[:, None] + 1e-3 + g.flatten(1).sum(1)[None] + 1e-3)
    mc = -(torch.einsum('nhw,mhw->nm', p2, g) + torch.einsum('nhw,mhw->nm', 1 - p2, 1 - g)) / (H * W)
    return 4 * dice + mc


print(f'libvkn cost matrix      : {timeit(lambda: a.cost_matrix(lo, cl, gt, lab)):.3f} ms')
print(f'torch ops cost matrix   : {timeit(torch_cost):.3f} ms')
print(f'libvkn assign (incl. D2H + C++ LSAP): {timeit(lambda: a.assign(lo, cl, gt, lab)):.3f} ms')

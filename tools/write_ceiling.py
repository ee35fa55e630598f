#!/usr/bin/env python3
"""GPU diagnostic: pure-write ceiling of the box (torch fill / hipMemset) next to k_upsample's 1.96 GB output."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import vkn_import  # noqa: E402

vkn = vkn_import.load()
dev = 'cuda:0'
m = torch.randn(8, 117, 128, 256, device=dev)
big = torch.empty(8, 117, 512, 1024, device=dev)
nbytes = big.numel() * 4


def timeit(fn, reps=10):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for name, fn in (('torch fill_', lambda: big.fill_(1.0)), ('torch zero_ (memset)', lambda: big.zero_()),
                 ('k_upsample x4', lambda: vkn.ops.upsample_bilinear(m, 4)),
                 ('torch interpolate x4', lambda: torch.nn.functional.interpolate(m, scale_factor=4, mode='bilinear', align_corners=False))):
    ms = timeit(fn)
    print(f'{name:24s} {ms*1e3:8.1f} us  {nbytes/ms/1e6:8.1f} GB/s written')

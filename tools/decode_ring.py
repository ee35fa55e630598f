#!/usr/bin/env python3
"""GPU diagnostic: decode ring depth (VKN_DECODE_RING = 3 | 4) x pixels per workgroup, interleaved repeats."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import vkn_import  # noqa: E402

vkn = vkn_import.load()
dev = 'cuda:0'
B, N, C, H, W = 8, 117, 256, 128, 256
x = torch.randn(B, C, H, W, device=dev)
k = torch.randn(B, N, C, device=dev)
hi, lo = vkn.ops.split_planes(k)
kb = torch.randn(B, N, device=dev)
out = torch.empty(B, N, H, W, device=dev)
ref = None
alg = B * H * W * (C * 4 + N * 4)


def timeit(reps=50):
    for _ in range(10):
        vkn.ops.mask_decode_planes(x, hi, lo, N, kb, out)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        vkn.ops.mask_decode_planes(x, hi, lo, N, kb, out)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for rnd in range(3):
    for ring in ('3', '4'):
        for ppw in ('512', '1024'):
            os.environ['VKN_DECODE_RING'] = ring
            os.environ['VKN_DECODE_PXWG'] = ppw
            ms = timeit()
            if ref is None:
                ref = out.clone()
            same = torch.equal(out, ref)
            print(f'round {rnd} ring={ring} px_per_wg={ppw}: {ms*1e3:6.1f} us {alg/ms/1e6:7.1f} GB/s  identical={same}')

#!/usr/bin/env python3
"""GPU diagnostic: decode kernel under tuning knobs (cache policy variants via VKN_DECODE_ABL, pixels per workgroup, XCD remap)."""
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
alg = B * H * W * (C * 4 + N * 4)


def timeit(fn, reps=40):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def run(tag):
    ms = timeit(lambda: vkn.ops.mask_decode_planes(x, hi, lo, N, kb, out))
    print(f'{tag:44s} {ms*1e3:7.1f} us  {alg/ms/1e6:7.1f} GB/s')


for rnd in range(2):
    for abl, name in (('0', 'sc0|nt loads, plain stores (default)'), ('5', 'default + nt stores'), ('4', 'plain loads'),
                      ('6', 'nt loads')):
        os.environ['VKN_DECODE_ABL'] = abl
        run(f'round {rnd} ABL={abl} {name}')
os.environ['VKN_DECODE_ABL'] = '0'
for ppw in (512, 1024):
    for xcd in (0, 1):
        os.environ['VKN_DECODE_PXWG'] = str(ppw)
        os.environ['VKN_DECODE_XCD'] = str(xcd)
        run(f'px_per_wg={ppw} xcd_remap={xcd}')

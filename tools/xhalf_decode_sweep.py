#!/usr/bin/env python3
"""Half-storage decode: pixels-per-workgroup sweep (debug library, VKN_DECODE_PXWG), B = 32, cfg2."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools'))
from perf_r02 import timeit  # noqa: E402

import vkn_import  # noqa: E402
vkn = vkn_import.load()
vkn._lib.build_debug()
vkn._lib.use_debug()
dev = torch.device('cuda', 0)
B, N, C, H, W = 32, 117, 256, 128, 256
P = H * W
x32 = torch.randn(B, C, H, W, device=dev)
kern = torch.randn(B, N, C, device=dev)
hi, lo = vkn.ops.split_planes(kern)
kb = torch.randn(B, N, device=dev)
out = torch.empty(B, N, H, W, device=dev)
ref = None
x = x32.to(torch.bfloat16)
for pair in (0, 1, 0, 1):
    os.environ['VKN_DECODE_XPAIR'] = str(pair)
    alg = B * P * (C * 2 + N * 4)
    t = timeit(lambda: vkn.ops.mask_decode_planes(x, hi, lo, N, kb, out), reps=30)
    same = True if ref is None else torch.equal(ref, out)
    ref = out.clone() if ref is None else ref
    print(f'x=bf16 paired 8-byte loads={pair}: {t:8.1f} us  {alg / t / 1e6:6.3f} TB/s  frac {alg / t / 8e6:.3f}  same bits: {same}', flush=True)
os.environ.pop('VKN_DECODE_XPAIR')
for nm, dt in (('bf16', torch.bfloat16),):
    x = x32.to(dt)
    alg = B * P * (C * x.element_size() + N * 4)
    for ppw in (0, 512, 1024, 2048, 4096, 8192, 16384):
        if ppw:
            os.environ['VKN_DECODE_PXWG'] = str(ppw)
        else:
            os.environ.pop('VKN_DECODE_PXWG', None)
        t = timeit(lambda: vkn.ops.mask_decode_planes(x, hi, lo, N, kb, out), reps=30)
        print(f'x={nm} px/wg={ppw or "policy"}: {t:8.1f} us  {alg / t / 1e6:6.3f} TB/s  frac {alg / t / 8e6:.3f}', flush=True)

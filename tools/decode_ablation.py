#!/usr/bin/env python3
"""GPU diagnostic: time the decode kernel alone (B=8 cfg2) under ablations / workgroup sizes, and list PMC counter names."""
import os
import subprocess
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


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


only = os.environ.get('VKN_ONLY')
if only:       # used under rocprofv3 --pmc: just run the kernel a few times
    for _ in range(5):
        vkn.ops.mask_decode_planes(x, hi, lo, N, kb, out)
    torch.cuda.synchronize()
    sys.exit(0)

for abl in ('0', '1', '2', '3', '4', '5', '6'):
    os.environ['VKN_DECODE_ABL'] = abl
    ms = timeit(lambda: vkn.ops.mask_decode_planes(x, hi, lo, N, kb, out))
    print(f'ABL={abl} ({["real", "no-mfma", "no-loads", "no-stores", "nt-loads", "nt-stores", "nt-both"][int(abl)]}): {ms*1e3:8.1f} us  -> {alg/ms/1e6:7.1f} GB/s algorithmic')
os.environ.pop('VKN_DECODE_ABL')
for ppw in (256, 512, 1024, 2048, 4096):
    os.environ['VKN_DECODE_PXWG'] = str(ppw)
    ms = timeit(lambda: vkn.ops.mask_decode_planes(x, hi, lo, N, kb, out))
    print(f'px_per_wg={ppw}: {ms*1e3:8.1f} us  -> {alg/ms/1e6:7.1f} GB/s')
os.environ.pop('VKN_DECODE_PXWG')
# row-stride sensitivity: same bytes, non-power-of-two P (channel rows no longer 128 KB apart)
for (h2, w2) in ((128, 250), (125, 256), (128, 264)):
    x2 = torch.randn(B, C, h2, w2, device=dev)
    out2 = torch.empty(B, N, h2, w2, device=dev)
    ms = timeit(lambda: vkn.ops.mask_decode_planes(x2, hi, lo, N, kb, out2))
    a2 = B * h2 * w2 * (C * 4 + N * 4)
    print(f'P={h2*w2} ({h2}x{w2}): {ms*1e3:8.1f} us  -> {a2/ms/1e6:7.1f} GB/s')
    m2 = torch.randn(B, N, h2, w2, device=dev)
    ms = timeit(lambda: vkn.ops.mask_gather(x2, m2))
    print(f'   gather+reduce P={h2*w2}: {ms*1e3:8.1f} us  -> {a2/ms/1e6:7.1f} GB/s')
# reference points: a plain copy of the same bytes, and torch's own fp32 bmm
y = torch.empty_like(x)
ms = timeit(lambda: y.copy_(x))
print(f'torch copy of x (read+write {2*x.numel()*4/1e6:.0f} MB): {ms*1e3:.1f} us -> {2*x.numel()*4/ms/1e6:.1f} GB/s')
ms = timeit(lambda: torch.bmm(k, x.reshape(B, C, -1)))
print(f'torch.bmm fp32 (rocBLAS) same op: {ms*1e3:.1f} us -> {alg/ms/1e6:.1f} GB/s algorithmic')
xr = torch.randn(B, N, H, W, device=dev)
ms = timeit(lambda: vkn.ops.mask_gather(x, xr))
print(f'gather+reduce: {ms*1e3:.1f} us -> {alg/ms/1e6:.1f} GB/s')

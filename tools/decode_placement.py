#!/usr/bin/env python3
"""GPU diagnostic: is the decode kernel's rate sensitive to (a) sustained load (clocks), (b) the relative placement of x / out,
(c) the values in x (randn vs the bench's hash generator)?"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import vkn_import  # noqa: E402

vkn = vkn_import.load()
dev = 'cuda:0'
B, N, C, H, W = 8, 117, 256, 128, 256
alg = B * H * W * (C * 4 + N * 4)
k = torch.randn(B, N, C, device=dev)
hi, lo = vkn.ops.split_planes(k)
kb = torch.randn(B, N, device=dev)


def timeit(x, out, reps=50, warm=10):
    for _ in range(warm):
        vkn.ops.mask_decode_planes(x, hi, lo, N, kb, out)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        vkn.ops.mask_decode_planes(x, hi, lo, N, kb, out)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    return f'{ms*1e3:7.1f} us {alg/ms/1e6:7.1f} GB/s'


x = torch.randn(B, C, H, W, device=dev)
out = torch.empty(B, N, H, W, device=dev)
print('cold, fresh allocations        ', timeit(x, out, warm=3))
print('again                          ', timeit(x, out))
t0 = time.time()
while time.time() - t0 < 3.0:
    for _ in range(100):
        vkn.ops.mask_decode_planes(x, hi, lo, N, kb, out)
    torch.cuda.synchronize()
print('after 3 s sustained decode     ', timeit(x, out))
# sustained mixed load like the bench (upsample = pure writes at full rate)
big = torch.empty(B, N, H * 4, W * 4, device=dev)
t0 = time.time()
while time.time() - t0 < 3.0:
    for _ in range(20):
        vkn.ops.upsample_bilinear(out, 4)
    torch.cuda.synchronize()
print('after 3 s sustained upsample   ', timeit(x, out))
del big
time.sleep(2.0)
print('after 2 s idle                 ', timeit(x, out, warm=3))

# placement: carve x and out from one arena at controlled relative offsets
arena = torch.empty((x.numel() + out.numel()) + (64 << 20), dtype=torch.float32, device=dev)
for off_kb in (0, 4, 64, 256, 1024, 2048 + 4, 4096 + 64 + 4):
    xo = 0
    oo = x.numel() + off_kb * 256
    xa = arena[xo:xo + x.numel()].view_as(x).copy_(x)
    oa = arena[oo:oo + out.numel()].view_as(out)
    print(f'arena, out = x_end + {off_kb:5d} KiB   ', timeit(xa, oa))

sys.path.insert(0, ROOT)
from oracle import synth  # noqa: E402  (diagnostic only: the bench's input generator)
xs = torch.from_numpy(synth.normalish((B, C, H, W), 11)).to(dev)
print('hash-generated x               ', timeit(xs, out))
print('zeros x                        ', timeit(torch.zeros_like(x), out))

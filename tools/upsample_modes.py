#!/usr/bin/env python3
"""GPU diagnostic: k_upsample_s variants (VKN_UPSAMPLE: 0 generic, 1 staged+nt, 2 staged+plain stores, 3 no loads, 4 stores only
nt, 5 stores only plain) against torch fill_, interleaved in one process (boxes differ by +-5 %)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import vkn_import  # noqa: E402

vkn = vkn_import.load()
dev = 'cuda:0'
B = int(os.environ.get("B", 8))
m = torch.randn(B, 117, 128, 256, device=dev)
big = torch.empty(B, 117, 512, 1024, device=dev)
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


for rnd in range(3):
    line = [f'fill {timeit(lambda: big.fill_(1.0))*1e3:6.1f}']
    for mode in sys.argv[1:] or ['0', '1', '2', '4', '5']:
        os.environ['VKN_UPSAMPLE'] = mode
        line.append(f'm{mode} {timeit(lambda: vkn.ops.upsample_bilinear(m, 4))*1e3:6.1f}')
    print('round', rnd, ' | '.join(line), 'us')

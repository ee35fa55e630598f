#!/usr/bin/env python3
"""GPU diagnostic: step time of the bench workload over the first seconds of a fresh process on a fresh box (clock / power ramp)."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import vkn_import  # noqa: E402

vkn = vkn_import.load()
dev = torch.device('cuda', 0)
head = bench.build_head(vkn, dev)
B = int(os.environ.get('B', 32))
x, pf, mp = bench.synth_inputs(B, dev, 0)
N, C = bench.CFG2['N'], bench.CFG2['C']
dims = head.mask_head[-1].make_dims(B, N, bench.CFG2['H'], bench.CFG2['W'])
packs = [h.stage_pack(dev) for h in head.mask_head]
pfr = pf.reshape(B, N, C)
t_start = time.perf_counter()
with torch.no_grad():
    while time.perf_counter() - t_start < float(os.environ.get('SECS', 14)):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(25):
            out = vkn.ops.head_forward(dims, packs, x, pfr, mp, None, 4)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(f't={time.perf_counter() - t_start:6.2f}s  {dt / 25 * 1e3:7.3f} ms/step', flush=True)

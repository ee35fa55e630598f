#!/usr/bin/env python3
"""GPU diagnostic: host time of one head_forward call vs its GPU time; and the same under a captured HIP graph."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import bench  # noqa: E402
import vkn_import  # noqa: E402

vkn = vkn_import.load()
dev = torch.device('cuda', 0)
head = bench.build_head(vkn, dev)
B = 8
x, pf, mp = bench.synth_inputs(B, dev, 0)
N, C = bench.CFG2['N'], bench.CFG2['C']
last = head.mask_head[-1]
dims = last.make_dims(B, N, bench.CFG2['H'], bench.CFG2['W'])
packs = [h.stage_pack(dev) for h in head.mask_head]
pfr = pf.reshape(B, N, C)


def call(up):
    return vkn.ops.head_forward(dims, packs, x, pfr, mp, None, up)


for up in (1, 4):
    for _ in range(3):
        call(up)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        call(up)
    t_host = (time.perf_counter() - t0) / 20
    torch.cuda.synchronize()
    t_all = (time.perf_counter() - t0) / 20
    print(f'up={up}: host enqueue {t_host*1e3:.3f} ms per call, wall (with final sync) {t_all*1e3:.3f} ms per call')

# HIP graph capture of the whole call
try:
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3):
            call(4)
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = call(4)
    torch.cuda.synchronize()
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        g.replay()
    torch.cuda.synchronize()
    print(f'graph replay up=4: {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms per call')
    ref = call(4)
    torch.cuda.synchronize()
    print('graph output equals eager:', all(torch.equal(a, b) for a, b in zip(out[:4], ref[:4])))
except Exception as e:  # noqa: BLE001
    print('graph capture failed:', repr(e)[:300])

#!/usr/bin/env python3
"""Is one `vkn_head_forward` call (S stages + link on the library's side stream + x4 upsample) capturable into a hipGraph by the
CALLER, and what does a replay buy at 1 / 2 / 8 frames per call, where the step is bound by ~50 kernel launches from the host?
usage (GPU box): python tools/graph_replay.py"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
import vkn_import
vkn = vkn_import.load()
dev = torch.device('cuda', 0)
head = bench.build_head(vkn, dev)
C2 = bench.CFG2
N, C = C2['N'], C2['C']
last = head.mask_head[-1]
packs = [h.stage_pack(dev) for h in head.mask_head]
first_prev = torch.zeros(1, N, C, device=dev)
for B in (1, 2, 8, 32):
    x, pf, mp = bench.synth_inputs(B, dev, 0)
    pf = pf.reshape(B, N, C)
    dims = last.make_dims(B, N, C2['H'], C2['W'])
    call = lambda: vkn.ops.head_forward(dims, packs, x, pf, mp, None, C2['up'], clip_first_prev=first_prev)
    with torch.no_grad():
        for _ in range(5):
            ref = call()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(50):
            call()
        torch.cuda.synchronize()
        eager = (time.perf_counter() - t0) / 50
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            out = call()
        for _ in range(5):
            g.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(50):
            g.replay()
        torch.cuda.synchronize()
        rep = (time.perf_counter() - t0) / 50
    same = all(torch.equal(a, b) for a, b in zip(out, ref) if torch.is_tensor(a))
    print(f'B={B:2d} frames per call: eager {eager * 1e3:7.3f} ms ({B / eager:8.1f} frames/s)   hipGraph replay {rep * 1e3:7.3f} ms '
          f'({B / rep:8.1f} frames/s)   outputs bit-identical: {same}', flush=True)

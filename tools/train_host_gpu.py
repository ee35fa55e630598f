#!/usr/bin/env python3
"""Is the training bench step host-bound or GPU-bound?  Runs K steps without synchronising: `enqueue` = wall time until the K-th step()
returned (the host's work), `drain` = the wait of the final synchronize (what the GPU still had queued).  drain ~ 0 -> the host is the
bottleneck; enqueue << total -> the GPU is.  Also the GPU time of one step measured with events around a synchronised step.
usage (GPU box): python tools/train_host_gpu.py [frames]"""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import vkn_import  # noqa: E402
from importlib import import_module  # noqa: E402

vkn = vkn_import.load()
vkn_dist = import_module('video_k_net_amd.dist')
device = torch.device('cuda', 0)
src = open(os.path.join(ROOT, 'bench.py')).read()
args = argparse.Namespace(frames=int(sys.argv[1]) if len(sys.argv) > 1 else 32, warmup=3, steps=10, no_chain_graphs=False, torch_chain=False, train_up=4)
body = src[src.index('def train_main('):src.index('    for _ in range(max(args.warmup, 3)):', src.index('def train_main('))]
ns = dict(bench.__dict__)
exec(body + '    return locals()\n', ns)
L = ns['train_main'](args, vkn, vkn_dist, device, 1, 0)
step = L['step']
for _ in range(8):
    step()
torch.cuda.synchronize()
K = 30
t0 = time.perf_counter()
for _ in range(K):
    step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f'{K} steps: enqueue {1e3 * (t1 - t0) / K:.3f} ms/step (host), drain after the last step {1e3 * (t2 - t1):.3f} ms, total {1e3 * (t2 - t0) / K:.3f} ms/step')
# one synchronised step: host time with an idle GPU (launch latency exposed) and GPU span by events
hs, gs = [], []
for _ in range(10):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(L['train_stream']):
        e0.record()
    h0 = time.perf_counter()
    step()
    h1 = time.perf_counter()
    with torch.cuda.stream(L['train_stream']):
        e1.record()
    torch.cuda.synchronize()
    hs.append(1e3 * (h1 - h0))
    gs.append(e0.elapsed_time(e1))
print(f'isolated step: host {sorted(hs)[len(hs) // 2]:.3f} ms, GPU span (first to last kernel) {sorted(gs)[len(gs) // 2]:.3f} ms')

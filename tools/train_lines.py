#!/usr/bin/env python3
"""Where the training bench step's GPU time and kernel launches come from, BY SOURCE LINE of this package (forward / loss / target
building) and by autograd node (backward): torch.profiler with stacks over a few steady-state steps.  Every kernel is attributed to
the innermost frame of video-k-net_amd / bench.py on the launching op's Python stack; kernels launched from the autograd engine's
thread go to the autograd node that ran them.   usage (GPU box): python tools/train_lines.py [top] [frames]"""
import argparse
import collections
import os
import sys

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
args = argparse.Namespace(frames=int(sys.argv[2]) if len(sys.argv) > 2 else 32, warmup=3, steps=10, no_chain_graphs=False, torch_chain=False, train_up=4)
body = src[src.index('def train_main('):src.index('    def step():', src.index('def train_main('))]
ns = dict(bench.__dict__)
exec(body + '    return locals()\n', ns)
L = ns['train_main'](args, vkn, vkn_dist, device, 1, 0)
head, reducer, opt, x, pf, mp, metas = L['head'], L['reducer'], L['opt'], L['x'], L['pf'], L['mp'], L['metas']
gt_masks, gt_labels, gt_sem_seg, gt_sem_cls, prev = L['gt_masks'], L['gt_labels'], L['gt_sem_seg'], L['gt_sem_cls'], L['prev']


def step():
    with torch.cuda.stream(L['train_stream']):
        reducer.zero_grad(set_to_none=True)
        x.grad = None
        out = head.forward_train_with_previous(x, pf, mp, None, metas, gt_masks, gt_labels, gt_sem_seg=gt_sem_seg, gt_sem_cls=gt_sem_cls,
                                               previous_obj_feats=prev)
        loss = sum(v for k, v in out[0].items() if 'loss' in k) + 1e-3 * (out[5] ** 2).mean()
        loss.backward()
        reducer.finalize()
        opt.step()


for _ in range(8):
    step()
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity  # noqa: E402

NST = 3
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True,
             experimental_config=torch._C._profiler._ExperimentalConfig(verbose=True)) as prof:
    for _ in range(NST):
        step()
    torch.cuda.synchronize()


def where(e):
    for s in (e.stack or []):
        if 'video-k-net_amd/' in s or 'bench.py' in s or 'train_lines.py' in s:
            f = s.split('video-k-net_amd/')[-1] if 'video-k-net_amd/' in s else s.split('/')[-1]
            return f.strip()
    p = e
    while p is not None:
        if p.name.startswith('autograd::engine::evaluate_function'):
            return 'BACKWARD ' + p.name.split(': ', 1)[-1]
        if p.name.startswith('Optimizer.step'):
            return 'OPT ' + p.name
        p = p.cpu_parent
    p, top = e, e
    while p is not None:
        top, p = p, p.cpu_parent
    return 'OTHER ' + top.name


agg = collections.defaultdict(lambda: [0, 0.0, collections.Counter()])
for e in prof.events():
    ks = getattr(e, 'kernels', None)
    if not ks:
        continue
    # only the op that launched directly: skip parents whose children carry the same kernels
    if any(getattr(c, 'kernels', None) for c in (e.cpu_children or [])):
        continue
    w = where(e)
    for k in ks:
        agg[w][0] += 1
        agg[w][1] += k.duration
        agg[w][2][k.name.replace('void ', '').split('(')[0].split('<')[0][-28:]] += 1
tot_n = sum(v[0] for v in agg.values())
tot_t = sum(v[1] for v in agg.values())
print(f'{tot_n / NST:.0f} kernels, {tot_t / NST:.0f} us of kernel time per step ({args.frames if args.frames != 32 else 4} frames)')
top = int(sys.argv[1]) if len(sys.argv) > 1 else 70
for w, (n, t, names) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
    kn = ', '.join(f'{c // NST}x {nm}' for nm, c in names.most_common(7))
    print(f'{t / NST:8.1f} us {n / NST:6.1f} k  {w[:70]:70s}  [{kn}]')

#!/usr/bin/env python3
"""Wall-clock split of the training bench step (synchronised after every section): forward_train, backward, all-reduce + SGD."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
src = open(os.path.join(ROOT, 'bench.py')).read()
# reuse bench.train_main's setup by executing it up to `def step():`
import types, argparse
import vkn_import
vkn = vkn_import.load()
from importlib import import_module
vkn_dist = import_module('video_k_net_amd.dist')
device = torch.device('cuda', 0)
args = argparse.Namespace(frames=int(sys.argv[1]) if len(sys.argv) > 1 else 32, warmup=3, steps=10)
body = src[src.index('def train_main('):src.index('    def step():', src.index('def train_main('))]
ns = dict(bench.__dict__)
exec(body + '    return locals()\n', ns)
L = ns['train_main'](args, vkn, vkn_dist, device, 1, 0)
head, reducer, opt, x, pf, mp, metas = L['head'], L['reducer'], L['opt'], L['x'], L['pf'], L['mp'], L['metas']
gt_masks, gt_labels, gt_sem_seg, gt_sem_cls, prev = L['gt_masks'], L['gt_labels'], L['gt_sem_seg'], L['gt_sem_cls'], L['prev']
B = L['B']
def sync():
    torch.cuda.synchronize(); return time.perf_counter()
tot = [0.0, 0.0, 0.0]
for it in range(13):
    reducer.zero_grad(set_to_none=not os.environ.get('ADDGRADS')); x.grad = None
    t0 = sync()
    out = head.forward_train_with_previous(x, pf, mp, None, metas, gt_masks, gt_labels, gt_sem_seg=gt_sem_seg, gt_sem_cls=gt_sem_cls,
                                           previous_obj_feats=prev)
    loss = sum(v for k, v in out[0].items() if 'loss' in k) + 1e-3 * (out[5] ** 2).mean()
    t1 = sync()
    loss.backward()
    t2 = sync()
    reducer.finalize(); opt.step()
    t3 = sync()
    if it >= 3:
        tot[0] += t1 - t0; tot[1] += t2 - t1; tot[2] += t3 - t2
n = 10
print(f'B={B} frames per step: forward_train {tot[0] / n * 1e3:.1f} ms, backward {tot[1] / n * 1e3:.1f} ms, all-reduce + SGD {tot[2] / n * 1e3:.1f} ms')

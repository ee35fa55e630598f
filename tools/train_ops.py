#!/usr/bin/env python3
"""Which torch ops the training bench step issues, by count and host time (steady state): torch.profiler over 5 steps after warm-up.
usage (GPU box): python tools/train_ops.py [top]"""
import os, sys, argparse
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
import vkn_import
from importlib import import_module
vkn = vkn_import.load()
vkn_dist = import_module('video_k_net_amd.dist')
device = torch.device('cuda', 0)
src = open(os.path.join(ROOT, 'bench.py')).read()
args = argparse.Namespace(frames=32, warmup=3, steps=10, no_chain_graphs=bool(os.environ.get('NOGRAPH')), torch_chain=bool(os.environ.get('TORCHCHAIN')))
body = src[src.index('def train_main('):src.index('    def step():', src.index('def train_main('))]
ns = dict(bench.__dict__)
exec(body + '    return locals()\n', ns)
L = ns['train_main'](args, vkn, vkn_dist, device, 1, 0)  # (chain graphs on unless NOGRAPH=1)
head, reducer, opt, x, pf, mp, metas = L['head'], L['reducer'], L['opt'], L['x'], L['pf'], L['mp'], L['metas']
gt_masks, gt_labels, gt_sem_seg, gt_sem_cls, prev = L['gt_masks'], L['gt_labels'], L['gt_sem_seg'], L['gt_sem_cls'], L['prev']


def step():
    if os.environ.get('DEFSTREAM'):
        return step_()
    with torch.cuda.stream(L['train_stream']):
        return step_()


def step_():
    reducer.zero_grad(set_to_none=not os.environ.get('ADDGRADS')); x.grad = None
    out = head.forward_train_with_previous(x, pf, mp, None, metas, gt_masks, gt_labels, gt_sem_seg=gt_sem_seg, gt_sem_cls=gt_sem_cls,
                                           previous_obj_feats=prev)
    loss = sum(v for k, v in out[0].items() if 'loss' in k) + 1e-3 * (out[5] ** 2).mean()
    loss.backward()
    reducer.finalize(); opt.step()


for _ in range(8):
    step()
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=bool(os.environ.get('SHAPES'))) as prof:
    for _ in range(5):
        step()
    torch.cuda.synchronize()
top = int(sys.argv[1]) if len(sys.argv) > 1 else 45
print(prof.key_averages(group_by_input_shape=bool(os.environ.get('SHAPES'))).table(sort_by=os.environ.get('SORT', 'self_cpu_time_total'), row_limit=top, max_name_column_width=60))
ev = prof.key_averages()
print('total op calls per step:', sum(e.count for e in ev) / 5)
print('kernel launches per step (hipLaunchKernel + ExtLaunch):', sum(e.count for e in ev if 'aunch' in e.key) / 5)
print('\nsynchronising ops per step:')
for e in ev:
    if any(k in e.key for k in ('item', '_local_scalar_dense', 'nonzero', 'is_nonzero', 'hipMemcpy', 'hipStreamSynchronize', 'hipDeviceSynchronize', 'hipEventSynchronize', '_to_copy', 'aten::to')):
        print(f'  {e.key:40s} {e.count / 5:8.1f} calls  {e.self_cpu_time_total / 5 / 1e3:8.3f} ms self cpu')
if os.environ.get('STACKS'):
    with profile(activities=[ProfilerActivity.CPU], with_stack=True) as p2:
        step()
        torch.cuda.synchronize()
    for e in p2.events():
        if e.name in ('aten::_local_scalar_dense', 'aten::nonzero') or 'hipMemcpyWithStream' in e.name:
            st = [s for s in (e.stack or []) if 'video-k-net_amd' in s or 'bench' in s or 'train_ops' in s][:3]
            print(e.name, '<-', ' | '.join(st))
if os.environ.get('CPROFILE'):
    import cProfile, pstats
    pr = cProfile.Profile()
    torch.cuda.synchronize()
    pr.enable()
    for _ in range(20):
        step()
    torch.cuda.synchronize()
    pr.disable()
    st = pstats.Stats(pr)
    st.sort_stats('cumulative').print_stats('video-k-net_amd|train_ops|torch/autograd/function|torch/cuda/graphs', 45)
    st.sort_stats('tottime').print_stats(25)
if os.environ.get('FINDSYNC'):
    import traceback
    for name in ('item', '__bool__', 'tolist', '__int__', '__float__', 'cpu', 'numpy', 'nonzero', '__index__'):
        orig = getattr(torch.Tensor, name)
        def wrap(self, *a, _orig=orig, _name=name, **k):
            if self.is_cuda:
                fr = [f for f in traceback.extract_stack()[:-1] if 'video-k-net_amd' in f.filename or 'train_ops' in f.filename or 'torch/optim' in f.filename][-2:]
                print('SYNC', _name, ' <- ', ' | '.join(f'{os.path.basename(f.filename)}:{f.lineno} {f.name}' for f in fr))
            return _orig(self, *a, **k)
        setattr(torch.Tensor, name, wrap)
    step()

"""A loop of chain forward + backward of ONE training stage (cfg3 shapes, 4 frames) — run under rocprofv3 to list what the device chain
launches (tools/chain_train_seq.py prints the sequence of the last iteration).   usage: chain_train_iter.py [kind] [iters] [torch]"""
import importlib
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
vkn = importlib.import_module('video-k-net_amd')
import test_gpu_chain_train as T  # noqa: E402

kind = sys.argv[1] if len(sys.argv) > 1 else 'video_ffn'
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
use_torch = len(sys.argv) > 3 and sys.argv[3] == 'torch'
over = {'video_update': dict(previous_link='update_dynamic_cov', previous_type='update'),
        'video_update_obj': dict(previous_link='link_atten', previous_type='update_obj')}.get(kind)
stage = T._head(vkn, kind != 'image', over)
B, N, C = 4, 117, 256
ins = [T._rand((B, N, C), 71, 3.0).requires_grad_(True), T._rand((B, N, C, 1, 1), 72).requires_grad_(True)]
if kind != 'image':
    ins.append(T._rand((B, N, C, 1, 1), 73).requires_grad_(True))
fn = (lambda *a: stage._chain_autograd(*a)) if use_torch else (lambda *a: vkn.chain_train.chain_forward(stage, *a))


def once():
    stage.zero_grad(set_to_none=True)
    outs = fn(*ins)
    gs = [torch.ones_like(o) * 1e-3 for o in outs if o is not None]
    torch.autograd.backward([o for o in outs if o is not None], gs)


for _ in range(5):
    once()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(iters):
    once()
torch.cuda.synchronize()
print(f'{kind} {"torch" if use_torch else "device"}: {(time.perf_counter() - t0) / iters * 1e3:.3f} ms per forward + backward (eager, host-paced)')

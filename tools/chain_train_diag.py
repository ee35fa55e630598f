"""Error of the device training chain and of the torch fp32 chain against the torch fp64 chain (outputs, input and parameter gradients)."""
import copy
import importlib
import sys
import os

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
vkn = importlib.import_module('video-k-net_amd')
import test_gpu_chain_train as T  # noqa: E402


def run(kind):
    over = {'video_update': dict(previous_link='update_dynamic_cov', previous_type='update'),
            'video_update_obj': dict(previous_link='link_atten', previous_type='update_obj')}.get(kind)
    stage = T._head(vkn, kind != 'image', over)
    B, N, C = 4, 117, 256
    ins = [T._rand((B, N, C), 71, 3.0), T._rand((B, N, C, 1, 1), 72)]
    if kind != 'image':
        ins.append(T._rand((B, N, C, 1, 1), 73))
    res = {}
    for mode in ('device', 'torch', 'fp64'):
        st = copy.deepcopy(stage).double() if mode == 'fp64' else stage
        st.zero_grad(set_to_none=True)
        args = [(t.double() if mode == 'fp64' else t).clone().requires_grad_(True) for t in ins]
        outs = vkn.chain_train.chain_forward(st, *args) if mode == 'device' else st._chain_autograd(*args)
        loss = 0
        for j, o in enumerate(outs):
            if o is not None:
                w = T._rand(tuple(o.shape), 80 + j, 1e-2)
                loss = loss + (o * (w.double() if mode == 'fp64' else w)).sum()
        loss.backward()
        res[mode] = ([o.detach().double() if o is not None else None for o in outs], [a.grad.double() for a in args],
                     {n: p.grad.double() for n, p in st.named_parameters() if p.grad is not None})
    ref = res['fp64']
    print(f'== {kind}')
    for mode in ('device', 'torch'):
        r = res[mode]
        eo = [T._rel(a, b) for a, b in zip(r[0], ref[0]) if a is not None]
        ei = [T._rel(a, b) for a, b in zip(r[1], ref[1])]
        ep = sorted(((T._rel(r[2][n], ref[2][n]), n) for n in ref[2]), reverse=True)
        print(f'  {mode:6s} outputs {max(eo):.2e}  input grads {" ".join("%.2e" % e for e in ei)}  worst params: '
              + ', '.join(f'{n} {e:.1e}' for e, n in ep[:4]))


if __name__ == '__main__':
    for k in sys.argv[1:] or ['image', 'video_ffn', 'video_update', 'video_update_obj']:
        run(k)

"""GPU: compare the gradients reaching every stage's outputs with the reference's (tools/diag/ref_stage_grads.py)."""
import sys, os
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import vkn_import
vkn = vkn_import.load()
from test_gpu_train import _train_case, DEV
name = sys.argv[1]
ref = dict(np.load(os.path.join(ROOT, 'tools/diag/_data', name + '_stage_grads.npz')))
g, case, head, (x, pf, mp, prev), (gt_masks, gt_labels, gt_sem_seg, gt_sem_cls) = _train_case(vkn, name)
if 'no_tail' in sys.argv: head.fused_tail = False
if 'torch_chain' in sys.argv:
    for st in head.mask_head: st.enable_device_chain(False)
if 'no_fused_losses' in sys.argv:
    for st in head.mask_head: st.fused_mask_losses = False
head.x_hub = 'no_hub' not in sys.argv
kept = []
orig = head._mask_forward
def wrapped(stage, *a, **kw):
    r = orig(stage, *a, **kw)
    for k in ('mask_preds', 'scaled_mask_preds', 'cls_score', 'object_feats'):
        if r.get(k) is not None and r[k].requires_grad:
            r[k].retain_grad()
    kept.append(r)
    return r
head._mask_forward = wrapped
xd = x.to(DEV).requires_grad_(True); pfd = pf.to(DEV).requires_grad_(True)
out = head.forward_train_with_previous(xd, pfd, mp.to(DEV), None, [dict() for _ in range(case['B'])], gt_masks, gt_labels, gt_sem_seg=gt_sem_seg,
                                       gt_sem_cls=gt_sem_cls, previous_obj_feats=prev.to(DEV))
total = sum(v for k, v in out[0].items() if 'loss' in k) + 0.01 * (out[5] ** 2).sum()
total.backward()
B, N = case['B'], case['N']
for s, r in enumerate(kept):
    for k in ('mask_preds', 'scaled_mask_preds', 'cls_score', 'object_feats'):
        for suffix, get in (('', lambda t: t.detach()), ('_grad', lambda t: t.grad)):
            key = f's{s}_{k}{suffix}'
            if key not in ref or r.get(k) is None or get(r[k]) is None:
                continue
            a, b = get(r[k]).cpu(), torch.from_numpy(ref[key])
            d = (a - b).abs().reshape(B, N, -1).amax(-1)
            sc = float(b.abs().max())
            w = int(d.argmax())
            print(f'{key:28s} max err {float(d.max()):.3e} / max {sc:.3e} = {float(d.max()) / max(sc, 1e-30):.2e} at (b,n)=({w // N},{w % N}); row(1,108) err {float(d[1, 108]):.3e} of rowmax {float(b.reshape(B, N, -1)[1, 108].abs().max()):.3e}')

"""Build container only (needs /root/reference): run a TRAIN_CASES golden case through the reference and dump the gradients that reach
every stage's outputs (d total / d mask_preds, scaled_mask_preds, cls_score, object_feats) -> /tmp/<case>_stage_grads.npz (a
diagnostic file, not a fixture)."""
import sys, os
sys.dont_write_bytecode = True
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'oracle'))
import gen_golden as G
import numpy as np, torch
from oracle import synth
name = sys.argv[1]
p = dict(G.TRAIN_CASES[name])
N, H, W, B, seed = (p.pop(k) for k in ('N', 'H', 'W', 'B', 'seed'))
p.pop('full_x', None)
cfg = G.head_cfg(**p)
cfg['train_cfg'] = [G.AttrDict(assigner=dict(type='MaskHungarianAssigner', cls_cost=dict(type='FocalLossCost', weight=2.0),
                                             dice_cost=dict(type='DiceCost', weight=4.0, pred_act=True),
                                             mask_cost=dict(type='MaskCost', weight=1.0, pred_act=True)),
                               sampler=dict(type='MaskPseudoSampler'), pos_weight=1) for _ in range(p['S'])]
head = G.build_head(cfg); head.train()
shapes = {k: tuple(v.shape) for k, v in head.state_dict().items()}
G.load_formula_weights(head, shapes, seed)
x, pf, mp = (torch.from_numpy(a) for a in synth.head_inputs(B, N, p['C'], H, W, seed))
x.requires_grad_(True); pf.requires_grad_(True)
tg = synth.train_targets(B, p['n_thing'], p['n_stuff'], H * p['up'], W * p['up'], seed)
t = lambda k: [torch.from_numpy(e[k]) for e in tg]
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
prev = torch.from_numpy(synth.normalish((B, N, p['C'], 1, 1), 99 + seed, 1.0))
out = head.forward_train_with_previous(x, pf, mp, None, [dict() for _ in range(B)], t('gt_masks'), t('gt_labels'), gt_sem_seg=t('gt_sem_seg'),
                                       gt_sem_cls=t('gt_sem_cls'), previous_obj_feats=prev)
losses, track = out[0], out[5]
total = sum(v for k, v in losses.items() if 'loss' in k) + 0.01 * (track ** 2).sum()
total.backward()
d = {}
for s, r in enumerate(kept):
    for k in ('mask_preds', 'scaled_mask_preds', 'cls_score', 'object_feats'):
        if r.get(k) is not None:
            d[f's{s}_{k}'] = r[k].detach().numpy()
            if r[k].grad is not None:
                d[f's{s}_{k}_grad'] = r[k].grad.numpy()
np.savez_compressed(f'/tmp/{name}_stage_grads.npz', **d)
print({k: v.shape for k, v in d.items()})

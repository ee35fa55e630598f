import os, sys
import numpy as np, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import vkn_import
vkn = vkn_import.load()
import test_gpu_vis as T
from oracle import synth
DEV='cuda:0'
name='vis_train_attnpos'
g, c, trk = T._build_train(vkn, name)
shapes = {k: tuple(v.shape) for k, v in trk.state_dict().items()}
trk.load_state_dict({k: torch.from_numpy(v) for k, v in synth.state_dict_like(shapes, c['seed'] + 1).items()}, strict=True)
trk = trk.to(DEV).train()
bs, nf, N, C, H, W = c['bs'], c['nf'], c['N'], c['C'], c['H'], c['W']
x, pf, mp = (torch.from_numpy(a) for a in synth.head_inputs(bs * nf, N, C, H, W, c['seed']))
x = x.reshape(bs, nf, C, H, W).to(DEV).requires_grad_(True)
obj = pf.reshape(bs, nf, N, C, 1, 1).to(DEV).requires_grad_(True)
masks = mp.reshape(bs, nf, N, H, W).to(DEV)
tg = synth.clip_targets(bs, nf, c['ncls'], H * c['up'], W * c['up'], c['seed'])
gt_masks = [[torch.from_numpy(m).to(DEV) for m in t['gt_masks']] for t in tg]
gt_labels = [torch.from_numpy(t['gt_labels']).to(DEV) for t in tg]
gt_ids = [torch.from_numpy(t['gt_instance_ids']).to(DEV) for t in tg]
assigned = []
costs = []
for a in trk.mask_assigner:
    orig = a.assign
    oc = a.cost_matrix
    def rec(*args, _orig=orig, **kw):
        r = _orig(*args, **kw)
        assigned.append(r[0].gt_inds.clone())
        return r
    def recc(*args, _oc=oc, **kw):
        r = _oc(*args, **kw); costs.append(r.clone()); return r
    a.assign = rec; a.cost_matrix = recc
omf = type(trk)._mask_forward
def mf(self, stage, x, of, mp):
    r = omf(self, stage, x, of, mp)
    if stage == 0:
        torch.set_printoptions(precision=6, linewidth=200)
        print('OUR cls', r['cls_score'][1, :3])
        print('OUR masks', r['mask_preds'][1, 0, :2, 0, :6])
        print('OUR obj', r['object_feats'][1, :2, :6, 0, 0])
    return r
type(trk)._mask_forward = mf
losses, feats = trk.forward_train(x, [[dict()] * nf for _ in range(bs)], None, masks, obj, gt_masks, gt_labels, gt_ids)
A = torch.stack(assigned).cpu().numpy()
print(A.shape, g['assigned'].shape)
for i in range(A.shape[0]):
    print(i, np.array_equal(A[i], g['assigned'][i]), A[i].tolist(), g['assigned'][i].tolist())
    if not np.array_equal(A[i], g['assigned'][i]):
        cm = costs[i].cpu().numpy()
        np.set_printoptions(precision=5, linewidth=200)
        print(cm.T)
for k, ref in zip(g['loss_keys'], g['loss_vals']):
    print(k, float(losses[k]), ref)

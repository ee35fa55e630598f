"""Diagnostic (GPU): where does the training step differ from a reference golden?  per-row errors of track / grad_pf, loss errors,
binarised-mask differences are not visible here (goldens do not store them) — rows are the kernels."""
import sys, os
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import vkn_import
vkn = vkn_import.load()
from test_gpu_train import _train_case, DEV

name = sys.argv[1] if len(sys.argv) > 1 else 'train_video_c256'
g, case, head, (x, pf, mp, prev), (gt_masks, gt_labels, gt_sem_seg, gt_sem_cls) = _train_case(vkn, name)
for arg in sys.argv[2:]:
    if arg == 'torch_chain':
        for st in head.mask_head: st.enable_device_chain(False)
    if arg == 'no_tail':
        head.fused_tail = False
    if arg == 'exact':
        for st in head.mask_head: st.vkn_flags = vkn.ops.FLAG_EXACT_GEMM
metas = [dict() for _ in range(case['B'])]
xd = x.to(DEV).requires_grad_(True); pfd = pf.to(DEV).requires_grad_(True)
out = head.forward_train_with_previous(xd, pfd, mp.to(DEV), None, metas, gt_masks, gt_labels, gt_sem_seg=gt_sem_seg, gt_sem_cls=gt_sem_cls,
                                       previous_obj_feats=prev.to(DEV))
losses, track = out[0], out[5]
for k, ref in zip(g['loss_keys'], g['loss_vals']):
    print(f'{k:16s} got {float(losses[str(k)]):.6f} ref {ref:.6f} rel {abs(float(losses[str(k)]) - ref) / max(1, abs(ref)):.2e}')
tr = torch.from_numpy(g['track'])
e = (track.detach().cpu() - tr).abs().reshape(case['B'], case['N'], -1).amax(-1)
print('track max err per frame', e.amax(1).tolist(), 'rows > 1e-3:', (e > 1e-3).sum(1).tolist(), 'rows > 1e-4:', (e > 1e-4).sum(1).tolist())
total = sum(v for k, v in losses.items() if 'loss' in k) + 0.01 * (track ** 2).sum()
print('total', float(total), float(g['total']))
total.backward()
def report(tag, got):
    got = got.detach().cpu()
    if tag in g:
        ref = torch.from_numpy(g[tag]); d = (got - ref).abs()
        print(f'{tag}: max err {float(d.max()):.3e} of max {float(ref.abs().max()):.3e}  rel-norm {float(d.double().norm() / ref.double().norm()):.3e}')
        if d.dim() >= 3 and d.shape[1] == case['N']:
            er = d.reshape(case['B'], case['N'], -1).amax(-1) / float(ref.abs().max())
            print('   rows with err > 2e-3 of max:', (er > 2e-3).sum(1).tolist(), ' worst rows', [torch.topk(er[b], 5).indices.tolist() for b in range(case['B'])],
                  [[round(float(v), 4) for v in torch.topk(er[b], 5).values] for b in range(case['B'])])
    else:
        idx, val = torch.from_numpy(g[tag + '_idx']), torch.from_numpy(g[tag + '_val'])
        d = (got.reshape(-1)[idx] - val).abs()
        print(f'{tag}: sampled max err {float(d.max()):.3e} of max {float(val.abs().max()):.3e}; norm got {float(got.double().norm()):.6e} ref {float(g[tag + "_norm"]):.6e}; '
              f'samples > 2e-3 of max: {int((d > 2e-3 * val.abs().max()).sum())} of {d.numel()}')
report('grad_x', xd.grad); report('grad_pf', pfd.grad)
named = dict(head.named_parameters())
for i, k in enumerate(g['grad_keys']):
    report(f'grad_{i}', named[str(k)].grad)
worst = 0
for k, ref in zip(g['all_keys'], g['all_gnorm']):
    p = named[str(k)]
    if ref >= 0:
        r = abs(float(p.grad.double().norm()) - ref) / max(ref, 1e-6); worst = max(worst, r)
        if r > 5e-3: print('  gnorm', k, r)
print('worst grad-norm rel err', worst)
if 'grad_pf_idx' in g:
    idx, val = torch.from_numpy(g['grad_pf_idx']), torch.from_numpy(g['grad_pf_val'])
    d = (pfd.grad.detach().cpu().reshape(-1)[idx] - val).abs() / val.abs().max()
    rows = idx // case['C']
    bad = d > 2e-3
    import collections
    cnt = collections.Counter(rows[bad].tolist())
    er = e.reshape(-1)
    print('grad_pf samples > 2e-3 by row: (row, n_bad, n_samples_on_row, worst, track_err_of_row)')
    for r, c in sorted(cnt.items()):
        on = rows == r
        print('  ', (r // case['N'], r % case['N']), c, int(on.sum()), round(float(d[on].max()), 4), f'{float(er[r]):.2e}')

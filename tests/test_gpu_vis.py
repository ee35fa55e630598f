"""GPU (-m gpu): the YouTube-VIS model family's heads (BASELINE cfg4) against goldens captured from the
reference's own `knet_vis` classes (oracle/gen_golden_vis.py): the per-frame roi head `KernelIterHeadVideo` (instance results +
the features it hands on) and the clip-level tracker `KernelFrameIterHeadVideo` / `KernelUpdateHeadVideo` (5-D gather over the
frames of a clip, clip-shared kernels for two stages, per-frame kernels for the last)."""
import os

import numpy as np
import pytest
import torch

from helpers import GOLDEN, maxabs
from oracle import synth

DEV = 'cuda:0'
VIS_FIELDS = ('C', 'heads', 'ffn', 'ncls', 'N', 'H', 'W', 'up', 'S', 'bs', 'nf', 'seed', 'kmax')


def _stage_cfg(vkn, typ, c, **extra):
    return vkn.configs.mask_head_cfg(False, C=c['C'], heads=c['heads'], ffn=c['ffn'], ncls=c['ncls'], n_thing=c['ncls'], n_stuff=0,
                                     up=c['up'], head_type=typ, loss_rank=None, **extra)


def _build(vkn, name):
    g = dict(np.load(os.path.join(GOLDEN, name + '.npz'), allow_pickle=False))
    c = dict(zip(VIS_FIELDS, (int(v) for v in g['case'])))
    test_cfg = dict(max_per_img=c['kmax'], mask_thr=0.5)
    roi = vkn.build_head(dict(type='KernelIterHeadVideo', num_stages=c['S'], stage_loss_weights=[1] * c['S'],
                              proposal_feature_channel=c['C'], num_thing_classes=c['ncls'], num_stuff_classes=0, num_proposals=c['N'],
                              test_cfg=test_cfg, mask_head=[_stage_cfg(vkn, 'KernelUpdateHead', c) for _ in range(c['S'])]))
    trk = vkn.build_head(dict(type='KernelFrameIterHeadVideo', num_proposals=c['N'], num_stages=3, assign_stages=2,
                              proposal_feature_channel=c['C'], stage_loss_weights=(1., 1., 1.), num_thing_classes=c['ncls'],
                              num_stuff_classes=0, test_cfg=test_cfg,
                              mask_head=_stage_cfg(vkn, 'KernelUpdateHeadVideo', c, num_proposals=c['N'])))
    return g, c, roi, trk


@pytest.mark.parametrize('name', ['vis_tiny', 'vis_cfg'])
def test_vis_heads_state_dict_matches_reference(vkn, name):
    """CPU: same keys and shapes as the reference's knet_vis modules (with_cls=False stages have no classification branch)."""
    g, c, roi, trk = _build(vkn, name)
    for tag, mod in (('roi', roi), ('trk', trk)):
        sd = mod.state_dict()
        assert sorted(sd) == list(g[tag + '_keys'])
        assert [str(tuple(sd[k].shape)) for k in sorted(sd)] == list(g[tag + '_shapes'])


def _unpack(bits, n, shape):
    return np.unpackbits(bits)[:n * shape[0] * shape[1]].reshape((n,) + shape).astype(bool)


@pytest.mark.gpu
@pytest.mark.parametrize('name', ['vis_tiny', 'vis_cfg'])
def test_vis_pipeline_vs_reference_golden(vkn, name):
    g, c, roi, trk = _build(vkn, name)
    for mod, seed in ((roi, c['seed']), (trk, c['seed'] + 1)):
        shapes = {k: tuple(v.shape) for k, v in mod.state_dict().items()}
        mod.load_state_dict({k: torch.from_numpy(v) for k, v in synth.state_dict_like(shapes, seed).items()}, strict=True)
    roi, trk = roi.to(DEV).eval(), trk.to(DEV).eval()
    B = c['bs'] * c['nf']
    x, pf, mp = (torch.from_numpy(a).to(DEV) for a in synth.head_inputs(B, c['N'], c['C'], c['H'], c['W'], c['seed']))
    H, W = c['H'], c['W']
    meta = dict(img_shape=(H * 8 - 4, W * 8 - 8, 3), batch_input_shape=(H * 8, W * 8), ori_shape=(H * 6, W * 6, 3))
    img_metas, ref_img_metas = [meta] * c['bs'], [[meta] * c['nf'] for _ in range(c['bs'])]
    with torch.no_grad():
        res, feats = roi.simple_test(x, pf, mp, None, img_metas, ref_img_metas, rescale=True)
        tres, tfeats = trk.simple_test(x=feats['x_feats'], img_metas=img_metas, ref_img_metas=ref_img_metas,
                                       cls_scores=feats['cls_scores'], masks=feats['masks'], obj_feats=feats['obj_feats'])
    # ---- per-frame roi head
    assert maxabs(feats['obj_feats'], g['roi_obj_feats']) < 2e-4 and maxabs(feats['cls_scores'], g['roi_cls_scores']) < 1e-5
    assert maxabs(feats['masks'], g['roi_masks']) < 1e-3
    oshape = (H * 6, W * 6)
    for i, (bbox_result, segm_result) in enumerate(res):
        scores = np.concatenate([bb[:, 4] for bb in bbox_result])
        labels = np.concatenate([np.full(len(bb), k) for k, bb in enumerate(bbox_result)])
        assert np.array_equal(labels, g[f'roi_labels{i}']) and np.max(np.abs(scores - g[f'roi_scores{i}'])) < 1e-5
        masks = np.stack([m for per in segm_result for m in per]).astype(bool)
        assert np.mean(masks != _unpack(g[f'roi_masks{i}'], int(g[f'roi_nmask{i}']), oshape)) < 2e-3
    # ---- clip-level tracker
    assert tuple(tfeats['obj_feats'].shape) == g['trk_obj_feats'].shape          # [bs, nf, N, C, 1, 1] after the per-frame stage
    assert maxabs(tfeats['obj_feats'], g['trk_obj_feats']) < 5e-4 and maxabs(tfeats['cls_scores'], g['trk_cls_scores']) < 1e-5
    assert maxabs(tfeats['masks'], g['trk_masks']) < 2e-3
    for b in range(c['bs']):
        for f in range(c['nf']):
            bbox_results, mask_results = tres[b][f]
            rows = np.concatenate([np.concatenate([r, np.full((len(r), 1), k)], axis=1) for k, r in enumerate(bbox_results)])
            ref = g[f'trk_rows{b}_{f}']
            assert rows.shape == ref.shape and np.array_equal(rows[:, [0, 6]], ref[:, [0, 6]])          # ids and labels
            assert np.max(np.abs(rows[:, 5] - ref[:, 5])) < 1e-5                                       # scores
            masks = np.stack([m for per in mask_results for m in per]).astype(bool)
            assert np.mean(masks != _unpack(g[f'trk_masks{b}_{f}'], int(g[f'trk_nmask{b}_{f}']), oshape)) < 2e-3

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
VIS_CASES = ['vis_tiny', 'vis_cfg', 'vis_attn_tiny', 'vis_attnpos_tiny', 'vis_attnpos_cfg']
VIS_FIELDS = ('C', 'heads', 'ffn', 'ncls', 'N', 'H', 'W', 'up', 'S', 'bs', 'nf', 'seed', 'kmax')


def _stage_cfg(vkn, typ, c, **extra):
    return vkn.configs.mask_head_cfg(False, C=c['C'], heads=c['heads'], ffn=c['ffn'], ncls=c['ncls'], n_thing=c['ncls'], n_stuff=0,
                                     up=c['up'], head_type=typ, loss_rank=None, **extra)


def _build(vkn, name):
    g = dict(np.load(os.path.join(GOLDEN, name + '.npz'), allow_pickle=False))
    c = dict(zip(VIS_FIELDS, (int(v) for v in g['case'])))
    merge = str(g['merge']) if 'merge' in g else 'mean'
    test_cfg = dict(max_per_img=c['kmax'], mask_thr=0.5)
    roi = vkn.build_head(dict(type='KernelIterHeadVideo', num_stages=c['S'], stage_loss_weights=[1] * c['S'],
                              proposal_feature_channel=c['C'], num_thing_classes=c['ncls'], num_stuff_classes=0, num_proposals=c['N'],
                              test_cfg=test_cfg, mask_head=[_stage_cfg(vkn, 'KernelUpdateHead', c) for _ in range(c['S'])]))
    trk = vkn.build_head(dict(type='KernelFrameIterHeadVideo', num_proposals=c['N'], num_stages=3, assign_stages=2,
                              proposal_feature_channel=c['C'], stage_loss_weights=(1., 1., 1.), num_thing_classes=c['ncls'],
                              num_stuff_classes=0, test_cfg=test_cfg, query_merge_method=merge,
                              mask_head=_stage_cfg(vkn, 'KernelUpdateHeadVideo', c, num_proposals=c['N'], query_merge_method=merge)))
    return g, c, roi, trk


@pytest.mark.parametrize('name', VIS_CASES)
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
@pytest.mark.parametrize('name', VIS_CASES)
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
    if 'trk_query_fusion' in g:   # the attention query merge alone, on the reference's own per-frame object features
        fused = trk._query_fusion(torch.from_numpy(g['roi_obj_feats']).to(DEV), c['bs'], c['nf'])
        assert maxabs(fused, g['trk_query_fusion']) < 2e-4
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


@pytest.mark.gpu
@pytest.mark.parametrize('case', [(2, 20, 64, 3, False), (1, 100, 256, 3, True), (2, 100, 256, 5, False), (1, 128, 128, 8, True),
                                              (3, 50, 256, 1, True),
                                              # the matrix-core attention (k_attn_mfma: <= 256 keys, head width 16 / 32 / 64), every key-block count
                                              (2, 10, 256, 2, True), (1, 60, 256, 2, False), (1, 90, 256, 2, True), (2, 128, 256, 2, True),
                                              (1, 33, 128, 3, True), (2, 7, 128, 1, False), (1, 117, 256, 1, False), (2, 100, 256, 2, False)])
def test_query_merge_op_vs_oracle(vkn, case):
    """`vkn_query_merge_f32` against the oracle's restatement: <= 256 keys (the LDS-staged attention kernel) and more (keys from
    global memory, scores in LDS), with and without the position table.  fp32 tolerance of the bf16x3 GEMMs."""
    from oracle.knet_oracle import query_merge
    B, N, C, F, with_pos = case[:5]
    heads = 8                                        # (the merge attention is built with 8 heads whatever the stages use: the reference's)
    g = torch.Generator().manual_seed(1000 + N + F)
    shapes = {'query_merge_attn.attn.in_proj_weight': (3 * C, C), 'query_merge_attn.attn.in_proj_bias': (3 * C,),
              'query_merge_attn.attn.out_proj.weight': (C, C), 'query_merge_attn.attn.out_proj.bias': (C,),
              'query_merge_norm.weight': (C,), 'query_merge_norm.bias': (C,),
              'query_merge_ffn.layers.0.0.weight': (8 * C, C), 'query_merge_ffn.layers.0.0.bias': (8 * C,),
              'query_merge_ffn.layers.1.weight': (C, 8 * C), 'query_merge_ffn.layers.1.bias': (C,),
              'query_merge_ffn_norm.weight': (C,), 'query_merge_ffn_norm.bias': (C,)}
    sd = {k: torch.from_numpy(v) for k, v in synth.state_dict_like(shapes, 77 + C).items()}
    query = torch.randn(B, N, C, generator=g)
    keys = torch.randn(B, F * N, C, generator=g)
    pos = torch.randn(N, C, generator=g) if with_pos else None
    ref = query_merge(sd, '', query, keys, pos, heads=heads)
    named = {k: v.to(DEV) for k, v in sd.items()}
    pack = vkn.ops.link_pack(named, torch.device(DEV), None, 'query_merge_attn', 'query_merge_norm', 'query_merge_ffn', 'query_merge_ffn_norm')
    dims = vkn.ops.make_dims(B, N, C, 8, 8, 8, 8 * C, 1, 0, 0)
    out = vkn.ops.query_merge(dims, pack, query.to(DEV), keys.to(DEV), pos.to(DEV) if with_pos else None)
    assert maxabs(out, ref) < 2e-4, maxabs(out, ref)
    with pytest.raises(ValueError):
        vkn.ops.query_merge(dims, pack, query.to(DEV), keys.to(DEV)[:, :-1], None)


VIS_TRAIN_FIELDS = ('C', 'heads', 'ffn', 'ncls', 'N', 'H', 'W', 'up', 'bs', 'nf', 'seed', 'mask_init')


def _build_train(vkn, name):
    g = dict(np.load(os.path.join(GOLDEN, name + '.npz'), allow_pickle=False))
    c = dict(zip(VIS_TRAIN_FIELDS, (int(v) for v in g['case'])))
    merge = str(g['merge'])
    train_cfg = dict(assigner=dict(type='MaskHungarianAssignerVideo', cls_cost=dict(type='FocalLossCost', weight=2.0),
                                   dice_cost=dict(type='DiceCost', weight=4.0, pred_act=True),
                                   mask_cost=dict(type='MaskCost', weight=1.0, pred_act=True)),
                     sampler=dict(type='MaskPseudoSampler'), pos_weight=1)
    trk = vkn.build_head(dict(type='KernelFrameIterHeadVideo', num_proposals=c['N'], num_stages=3, assign_stages=2,
                              proposal_feature_channel=c['C'], stage_loss_weights=(1., 1., 1.), num_thing_classes=c['ncls'],
                              num_stuff_classes=0, train_cfg=train_cfg, query_merge_method=merge, with_mask_init=bool(c['mask_init']),
                              mask_head=_stage_cfg(vkn, 'KernelUpdateHeadVideo', c, num_proposals=c['N'], query_merge_method=merge)))
    return g, c, trk


@pytest.mark.parametrize('name', ['vis_train_tiny', 'vis_train_attnpos'])
def test_vis_train_state_dict_matches_reference(vkn, name):
    g, c, trk = _build_train(vkn, name)
    sd = trk.state_dict()
    assert sorted(sd) == list(g['keys']) and [str(tuple(sd[k].shape)) for k in sorted(sd)] == list(g['shapes'])


@pytest.mark.gpu
def test_vis_roi_head_forward_train_is_the_stage_loop_over_flattened_clips(vkn):
    """`KernelIterHeadVideo.forward_train` (knet_vis/tracker/kernel_iter_head.py:139-242, the per-frame roi head of the VIS model):
    clip-shaped ground truth (`gt_masks[clip][frame]`, `(frame, label)` rows) flattened onto the image head's stage loop — same
    losses as the loop called on the flat lists — and the clip-shaped `features` it hands to the clip-level tracker head."""
    c = dict(C=32, heads=4, ffn=64, ncls=5, N=10, H=8, W=12, up=2, S=2)
    train_cfg = [dict(assigner=dict(type='MaskHungarianAssigner', cls_cost=dict(type='FocalLossCost', weight=2.0),
                                    dice_cost=dict(type='DiceCost', weight=4.0, pred_act=True),
                                    mask_cost=dict(type='MaskCost', weight=1.0, pred_act=True)),
                      sampler=dict(type='MaskPseudoSampler'), pos_weight=1) for _ in range(c['S'])]
    roi = vkn.build_head(dict(type='KernelIterHeadVideo', num_stages=c['S'], stage_loss_weights=[1] * c['S'], assign_stages=c['S'],
                              proposal_feature_channel=c['C'], num_thing_classes=c['ncls'], num_stuff_classes=0, num_proposals=c['N'],
                              train_cfg=train_cfg, mask_head=[_stage_cfg(vkn, 'KernelUpdateHead', c) for _ in range(c['S'])]))
    shapes = {k: tuple(v.shape) for k, v in roi.state_dict().items()}
    roi.load_state_dict({k: torch.from_numpy(v) for k, v in synth.state_dict_like(shapes, 91).items()}, strict=True)
    roi = roi.to(DEV).train()
    bs, nf = 2, 3
    x, pf, mp = (torch.from_numpy(a).to(DEV) for a in synth.head_inputs(bs * nf, c['N'], c['C'], c['H'], c['W'], 17))
    tg = synth.clip_targets(bs, nf, c['ncls'], c['H'] * c['up'], c['W'] * c['up'], 17)
    gt_masks = [[torch.from_numpy(m).to(DEV) for m in t['gt_masks']] for t in tg]
    gt_labels = [torch.from_numpy(t['gt_labels']).to(DEV) for t in tg]
    metas = [[dict() for _ in range(nf)] for _ in range(bs)]
    losses, feats = roi.forward_train(x, pf, mp, None, metas, gt_masks, gt_labels)
    flat_masks = [gt_masks[i][j] for i in range(bs) for j in range(nf)]
    flat_labels = [gt_labels[i][:, 1][gt_labels[i][:, 0] == j] for i in range(bs) for j in range(nf)]
    want, last = roi._train_stages(x, pf, mp, None, [dict()] * (bs * nf), flat_masks, flat_labels)
    assert sorted(losses) == sorted(want) and all(torch.equal(losses[k], want[k]) for k in want)
    assert tuple(feats['obj_feats'].shape) == (bs, nf, c['N'], c['C'], 1, 1) and tuple(feats['x_feats'].shape) == (bs, nf, c['C'], c['H'], c['W'])
    assert torch.equal(feats['masks'].reshape(bs * nf, c['N'], c['H'], c['W']), last['mask_preds'])
    assert tuple(feats['cls_scores'].shape) == (bs, nf, c['N'], c['ncls'])
    got = vkn.HEADS.get('VideoKernelIterHead').get_masked_feature(roi, x, mp)     # (the video panoptic head's helper: the HIP gather)
    assert torch.equal(got, vkn.ops.mask_gather(x, mp, 0.5)[0])


@pytest.mark.gpu
def test_vis_rpn_and_roi_training_vs_reference_golden(vkn):
    """The VIS model's rpn_head -> roi_head training hand-over against the REFERENCE's own classes (oracle/gen_golden_vis.py:
    run_rpn_roi_train): `ConvKernelHeadVideo.forward_train` behind a pass-through neck, then `KernelIterHeadVideo.forward_train` on
    its outputs, clip-shaped ground truth — both loss dicts 1e-4 relative, every Hungarian assignment bit-exact, the clip-shaped
    features the tracker head receives."""
    g = dict(np.load(os.path.join(GOLDEN, 'vis_rpn_train.npz'), allow_pickle=False))
    C, heads, ffn, ncls, nprop, H, W, up, S, bs, nf, seed = (int(v) for v in g['case'])
    c = dict(C=C, heads=heads, ffn=ffn, ncls=ncls, up=up)
    assign = dict(type='MaskHungarianAssigner', cls_cost=dict(type='FocalLossCost', weight=2.0),
                  dice_cost=dict(type='DiceCost', weight=4.0, pred_act=True), mask_cost=dict(type='MaskCost', weight=1.0, pred_act=True))
    tc = dict(assigner=assign, sampler=dict(type='MaskPseudoSampler'), pos_weight=1)
    rpn = vkn.build_head(dict(type='ConvKernelHeadVideo', num_proposals=nprop, in_channels=C, out_channels=C, num_loc_convs=0,
                              num_seg_convs=0, localization_fpn=None, conv_kernel_size=1, semantic_fpn=True, num_classes=ncls,
                              use_binary=True, proposal_feats_with_obj=True, feat_downsample_stride=up, feat_refine=False,
                              num_thing_classes=ncls, num_stuff_classes=0, cat_stuff_mask=False,
                              loss_seg=dict(type='FocalLoss', use_sigmoid=True, gamma=2.0, alpha=0.25, loss_weight=1.0),
                              loss_mask=dict(type='CrossEntropyLoss', use_sigmoid=True, loss_weight=1.0),
                              loss_dice=dict(type='DiceLoss', loss_weight=4.0), train_cfg=tc))
    roi = vkn.build_head(dict(type='KernelIterHeadVideo', num_stages=S, stage_loss_weights=[1] * S, assign_stages=S,
                              proposal_feature_channel=C, num_thing_classes=ncls, num_stuff_classes=0, num_proposals=nprop,
                              train_cfg=[tc for _ in range(S)], mask_head=[_stage_cfg(vkn, 'KernelUpdateHead', c) for _ in range(S)]))
    assert sorted(rpn.state_dict()) == list(g['rpn_keys_sd'])
    for m, sd_seed in ((rpn, seed), (roi, seed + 1)):
        shapes = {k: tuple(v.shape) for k, v in m.state_dict().items()}
        m.load_state_dict({k: torch.from_numpy(v) for k, v in synth.state_dict_like(shapes, sd_seed).items()}, strict=True)
        m.to(DEV).train()
    rpn._upstream_feats = lambda img: img
    F_ = bs * nf
    loc = torch.from_numpy(synth.uniform((F_, C, H, W), seed + 2, -1.0, 1.0)).to(DEV).requires_grad_(True)
    sem = torch.from_numpy(synth.uniform((F_, C, H, W), seed + 3, -1.0, 1.0)).to(DEV).requires_grad_(True)
    tg = synth.clip_targets(bs, nf, ncls, H * up, W * up, seed)
    gt_masks = [[torch.from_numpy(m).to(DEV) for m in t['gt_masks']] for t in tg]
    gt_labels = [torch.from_numpy(t['gt_labels']).to(DEV) for t in tg]
    metas = [[dict() for _ in range(nf)] for _ in range(bs)]
    assigned = []

    def hook(a):
        for meth in ('assign', 'assign_batch', 'assign_batch_lowres'):
            if hasattr(a, meth):
                orig = getattr(a, meth)

                def rec(*args, _orig=orig, **kw):
                    r = _orig(*args, **kw)
                    assigned.extend(x.gt_inds.clone() for x in (r if isinstance(r, list) else [r]))
                    return r
                setattr(a, meth, rec)
    hook(rpn.assigner)
    for a in roi.mask_assigner:
        hook(a)
    rl, prop, x_feats, masks, cls = rpn.forward_train((loc, sem), [dict()] * bs, metas, gt_masks, gt_labels)
    n_rpn = len(assigned)
    ll, feats = roi.forward_train(x_feats, prop, masks, cls, metas, gt_masks, gt_labels)
    assert sorted(rl) == list(g['rpn_keys']) and sorted(ll) == list(g['roi_keys'])
    got = torch.stack(assigned).cpu().numpy()
    assert np.array_equal(got[:n_rpn], g['assigned_rpn']) and np.array_equal(got[n_rpn:], g['assigned_roi'])
    for keys, vals, d in ((g['rpn_keys'], g['rpn_vals'], rl), (g['roi_keys'], g['roi_vals'], ll)):
        for k, ref in zip(keys, vals):
            assert abs(float(d[str(k)].detach()) - ref) < 1e-4 * max(1.0, abs(ref)), (str(k), float(d[str(k)].detach()), ref)
    assert maxabs(prop, g['proposal_feats']) < 1e-3 * (1 + float(np.abs(g['proposal_feats']).max()))
    assert maxabs(feats['obj_feats'], g['feat_obj']) < 1e-3 and maxabs(feats['cls_scores'], g['feat_cls']) < 1e-3
    rs = feats['masks'].detach().double().sum(dim=(-1, -2)).cpu().numpy()
    assert np.abs(rs - g['feat_mask_rowsum']).max() < 2e-3 * max(1.0, float(np.abs(g['feat_mask_rowsum']).max()))
    total = sum(v for k, v in rl.items() if 'loss' in k) + sum(v for k, v in ll.items() if 'loss' in k)
    assert abs(float(total.detach()) - float(g['total'])) < 1e-4 * abs(float(g['total']))
    total.backward()
    assert abs(float(loc.grad.double().norm()) - float(g['grad_loc_norm'])) < 5e-3 * float(g['grad_loc_norm'])
    assert abs(float(sem.grad.double().norm()) - float(g['grad_sem_norm'])) < 5e-3 * float(g['grad_sem_norm'])


def test_clip_instances_of_the_video_assigner(vkn):
    """CPU: the clip-instance table (instances in ascending id order; a frame where an instance is absent is a zero mask; one
    label per instance) against a direct per-instance / per-frame construction."""
    tg = synth.clip_targets(3, 4, 5, 6, 10, 7)
    for t in tg:
        masks = [torch.from_numpy(m) for m in t['gt_masks']]
        clip, labels, inst = vkn.MaskHungarianAssignerVideo.clip_instances(4, masks, torch.from_numpy(t['gt_labels']),
                                                                           torch.from_numpy(t['gt_instance_ids']))
        ids, lab = t['gt_instance_ids'], t['gt_labels']
        assert list(inst) == sorted(set(ids[:, 1].tolist())) and clip.shape == (len(inst), 4, 6, 10)
        for gi, iid in enumerate(inst):
            for f in range(4):
                rows = ids[ids[:, 0] == f, 1]
                hit = np.nonzero(rows == iid)[0]
                want = masks[f][hit[0]] if len(hit) else torch.zeros(6, 10)
                assert torch.equal(clip[gi, f], want)
                if len(hit):
                    assert int(labels[gi]) == int(lab[lab[:, 0] == f, 1][hit[0]])
    with pytest.raises(ValueError):
        bad = tg[0]['gt_labels'].copy()
        bad[0, 1] = (bad[0, 1] + 1) % 5
        first = tg[0]['gt_instance_ids'][0, 1]
        if (tg[0]['gt_instance_ids'][:, 1] == first).sum() < 2:
            raise ValueError('instance visible once: nothing to clash with')
        vkn.MaskHungarianAssignerVideo.clip_instances(4, [torch.from_numpy(m) for m in tg[0]['gt_masks']], torch.from_numpy(bad),
                                                      torch.from_numpy(tg[0]['gt_instance_ids']))


@pytest.mark.gpu
@pytest.mark.parametrize('name', ['vis_train_tiny', 'vis_train_attnpos'])
def test_vis_clip_training_vs_reference_golden(vkn, name):
    """`KernelFrameIterHeadVideo.forward_train` (clip-level assignment with MaskHungarianAssignerVideo, per-stage losses, gradients)
    against the reference's own forward_train: assignments bit-exact, losses 1e-4 relative, gradients 2e-3 of the tensor's max."""
    g, c, trk = _build_train(vkn, name)
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
    for a in trk.mask_assigner:
        orig = a.assign

        def rec(*args, _orig=orig, **kw):
            r = _orig(*args, **kw)
            assigned.append(r[0].gt_inds.clone())
            return r
        a.assign = rec
    losses, feats = trk.forward_train(x, [[dict()] * nf for _ in range(bs)], None, masks, obj, gt_masks, gt_labels, gt_ids)
    assert sorted(losses) == list(g['loss_keys'])
    assert np.array_equal(torch.stack(assigned).cpu().numpy(), g['assigned']), 'clip-level Hungarian assignments must be bit-exact'
    for k, ref in zip(g['loss_keys'], g['loss_vals']):
        assert abs(float(losses[k].detach()) - ref) < 1e-4 * max(1.0, abs(ref)), (k, float(losses[k].detach()), ref)
    assert feats['cls_scores'] is None                       # the last (per-frame) stage has no classification branch
    assert maxabs(feats['obj_feats'], g['feat_obj']) < 1e-3 and maxabs(feats['masks'], g['feat_masks']) < 2e-3
    total = sum(v for k, v in losses.items() if 'loss' in k) + 0.01 * (feats['obj_feats'] ** 2).sum()
    assert abs(float(total.detach()) - float(g['total'])) < 1e-4 * abs(float(g['total']))
    total.backward()

    def close(got, ref, tag, tol=2e-3):
        ref = torch.from_numpy(ref)
        assert maxabs(got.detach().cpu(), ref) < tol * max(float(ref.abs().max()), 1e-12), tag
    close(x.grad, g['grad_x'], 'grad_x')
    close(obj.grad, g['grad_obj'], 'grad_obj')
    named = dict(trk.named_parameters())
    for i, k in enumerate(g['grad_keys']):
        close(named[str(k)].grad, g[f'grad_{i}'], str(k))
    for k, ref in zip(g['all_keys'], g['all_gnorm']):       # every parameter's gradient norm
        p = named[str(k)]
        if ref < 0:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, str(k)
        else:
            assert p.grad is not None and abs(float(p.grad.double().norm()) - ref) < 5e-3 * max(ref, 1e-6), (str(k), ref)

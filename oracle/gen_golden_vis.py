#!/usr/bin/env python3
"""Goldens of the YouTube-VIS model family (`knet_vis/`, BASELINE cfg4) from the REFERENCE ITSELF — build container only.

A separate process from oracle/gen_golden.py: `knet_vis` registers heads under the SAME registry names as `knet`
(`KernelUpdateHead`, ...).  The reference's files are imported unmodified through oracle/standins/ (plus the `mmtrack.transform`
and `mmdet.datasets.coco_panoptic` stand-ins).  Only outputs are stored.

    python oracle/gen_golden_vis.py        # writes tests/golden/vis_*.npz

  vis_train_* `forward_train` of the tracker head: clip-level assignment (MaskHungarianAssignerVideo), losses, gradients
  vis_attn_* the same pipeline with query_merge_method = 'attention' / 'attention_pos' in the tracker and its clip-level stages
  vis_rpn_train  `ConvKernelHeadVideo.forward_train` (knet_vis/tracker/kernel_head.py:267-334) and `KernelIterHeadVideo.forward_train`
             (knet_vis/tracker/kernel_iter_head.py:139-242) on the heads it feeds: clip-shaped ground truth, losses, assignments
  vis_tiny   KernelIterHeadVideo (per-frame roi head, instance results + features) -> KernelFrameIterHeadVideo (clip-level tracker:
             query fusion 'mean', 3 stages with assign_stages = 2: two clip-level `with_cls` stages, one per-frame stage)
"""
import os
import sys

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = os.environ.get('VKN_REFERENCE', '/root/reference')
sys.path.insert(0, os.path.join(HERE, 'standins'))
sys.path.insert(1, REF)
sys.path.insert(2, ROOT)

import copy  # noqa: E402

import numpy as np  # noqa: E402
import torch  # noqa: E402

from oracle import synth  # noqa: E402

import knet_vis.kernel_updator  # noqa: E402,F401
import knet_vis.det.kernel_update_head  # noqa: E402,F401
import knet_vis.tracker.kernel_iter_head  # noqa: E402,F401
import knet_vis.tracker.kernel_update_head  # noqa: E402,F401
import knet_vis.tracker.kernel_frame_iter_head  # noqa: E402,F401
import knet_vis.tracker.kernel_head  # noqa: E402,F401  (ConvKernelHeadVideo: the VIS models' rpn_head)
import knet_vis.det.mask_hungarian_assigner  # noqa: E402,F401
import knet_vis.det.mask_pseudo_sampler  # noqa: E402,F401
import knet.cross_entropy_loss  # noqa: E402,F401
from mmdet.models.builder import build_head  # noqa: E402

OUT = os.environ.get('VKN_GOLDEN_OUT', os.path.join(ROOT, 'tests', 'golden'))   # tests/test_golden_regen.py regenerates into a tmp dir


class AttrDict(dict):
    __getattr__ = dict.__getitem__


def stage_cfg(typ, C, heads, ffn, ncls, up, **extra):
    d = dict(type=typ, num_classes=ncls, num_thing_classes=ncls, num_stuff_classes=0, num_ffn_fcs=2, num_heads=heads, num_cls_fcs=1,
             num_mask_fcs=1, feedforward_channels=ffn, in_channels=C, out_channels=C, dropout=0.0, mask_thr=0.5, conv_kernel_size=1,
             mask_upsample_stride=up, ffn_act_cfg=dict(type='ReLU', inplace=True), with_ffn=True,
             feat_transform_cfg=dict(conv_cfg=dict(type='Conv2d'), act_cfg=None),
             kernel_updator_cfg=dict(type='KernelUpdator', in_channels=C, feat_channels=C, out_channels=C, input_feat_shape=3,
                                     act_cfg=dict(type='ReLU', inplace=True), norm_cfg=dict(type='LN')),
             loss_mask=dict(type='CrossEntropyLoss', use_sigmoid=True, loss_weight=1.0), loss_dice=dict(type='DiceLoss', loss_weight=4.0),
             loss_cls=dict(type='FocalLoss', use_sigmoid=True, gamma=2.0, alpha=0.25, loss_weight=2.0))
    d.update(extra)
    return d


def pack_masks(mask_results):
    masks = [m for per_cls in mask_results for m in per_cls]
    return (np.packbits(np.stack(masks).astype(bool)), len(masks)) if masks else (np.zeros(0, np.uint8), 0)


def run(name, C, heads, ffn, ncls, N, H, W, up, S, bs, nf, seed, kmax, merge='mean'):
    test_cfg = AttrDict(max_per_img=kmax, mask_thr=0.5)
    roi = build_head(dict(type='KernelIterHeadVideo', num_stages=S, stage_loss_weights=[1] * S, proposal_feature_channel=C,
                          num_thing_classes=ncls, num_stuff_classes=0, num_proposals=N, test_cfg=test_cfg,
                          mask_head=[stage_cfg('KernelUpdateHead', C, heads, ffn, ncls, up) for _ in range(S)]))
    trk = build_head(dict(type='KernelFrameIterHeadVideo', num_proposals=N, num_stages=3, assign_stages=2, proposal_feature_channel=C,
                          stage_loss_weights=(1., 1., 1.), num_thing_classes=ncls, num_stuff_classes=0, test_cfg=test_cfg,
                          query_merge_method=merge,
                          mask_head=stage_cfg('KernelUpdateHeadVideo', C, heads, ffn, ncls, up, num_proposals=N, query_merge_method=merge)))
    roi.eval()
    trk.eval()
    out = dict(case=np.array([C, heads, ffn, ncls, N, H, W, up, S, bs, nf, seed, kmax], dtype=np.int64))
    if merge != 'mean':
        out['merge'] = np.array(merge)
    for tag, mod, sd_seed in (('roi', roi, seed), ('trk', trk, seed + 1)):
        shapes = {k: tuple(v.shape) for k, v in mod.state_dict().items()}
        mod.load_state_dict({k: torch.from_numpy(v) for k, v in synth.state_dict_like(shapes, sd_seed).items()}, strict=True)
        out[tag + '_keys'] = np.array(sorted(shapes))
        out[tag + '_shapes'] = np.array([str(shapes[k]) for k in sorted(shapes)])
    B = bs * nf
    x, pf, mp = (torch.from_numpy(a) for a in synth.head_inputs(B, N, C, H, W, seed))
    meta = dict(img_shape=(H * 8 - 4, W * 8 - 8, 3), batch_input_shape=(H * 8, W * 8), ori_shape=(H * 6, W * 6, 3))
    img_metas = [meta] * bs
    ref_img_metas = [[meta] * nf for _ in range(bs)]
    with torch.no_grad():
        res, feats = roi.simple_test(x, pf, mp, None, img_metas, ref_img_metas, rescale=True)
        tres, tfeats = trk.simple_test(x=feats['x_feats'], img_metas=img_metas, ref_img_metas=ref_img_metas, cls_scores=feats['cls_scores'],
                                       masks=feats['masks'], obj_feats=feats['obj_feats'])
    if merge != 'mean':   # the fused clip-level kernels alone (pins oracle.query_merge; tests/test_oracle_golden.py)
        with torch.no_grad():
            out['trk_query_fusion'] = trk._query_fusion(feats['obj_feats'], bs, nf).numpy()
    for k in ('obj_feats', 'cls_scores', 'masks'):
        out['roi_' + k] = feats[k].numpy()
        out['trk_' + k] = tfeats[k].numpy()
    for i, (bbox_result, segm_result) in enumerate(res):
        out[f'roi_scores{i}'] = np.concatenate([bb[:, 4] for bb in bbox_result])
        out[f'roi_labels{i}'] = np.concatenate([np.full(len(bb), c) for c, bb in enumerate(bbox_result)]).astype(np.int64)
        out[f'roi_masks{i}'], out[f'roi_nmask{i}'] = pack_masks(segm_result)
    for b in range(bs):
        for f in range(nf):
            bbox_results, mask_results = tres[b][f]
            out[f'trk_rows{b}_{f}'] = np.concatenate([np.concatenate([r, np.full((len(r), 1), c)], axis=1) for c, r in enumerate(bbox_results)])
            out[f'trk_masks{b}_{f}'], out[f'trk_nmask{b}_{f}'] = pack_masks(mask_results)
    np.savez_compressed(os.path.join(OUT, name + '.npz'), **out)
    print(f'{name}: ok  roi instances/frame = {[int(out[f"roi_nmask{i}"]) for i in range(B)]}, tracker instances = {int(out["trk_nmask0_0"])}')


TRAIN_GRAD_KEYS = ('mask_head.0.kernel_update_conv.dynamic_layer.weight', 'mask_head.0.attention.attn.in_proj_weight',
                   'mask_head.0.ffn.layers.1.weight', 'mask_head.0.fc_cls.weight', 'mask_head.0.feat_transform.conv.weight',
                   'mask_head.1.fc_mask.weight', 'mask_head.2.kernel_update_conv.fc_layer.weight', 'mask_head.2.fc_mask.bias',
                   'init_query.weight', 'query_pos.weight', 'query_merge_attn.attn.in_proj_weight', 'query_merge_ffn.layers.0.0.weight',
                   'mask_head.0.query_merge_attn.attn.out_proj.weight', 'mask_head.1.query_merge_ffn_norm.weight', 'fc_mask.weight')


def run_train(name, C, heads, ffn, ncls, N, H, W, up, bs, nf, seed, merge='mean', mask_init=False):
    """`KernelFrameIterHeadVideo.forward_train` of the reference (knet_vis/tracker/kernel_frame_iter_head.py:182-312) with the shipped
    `train_cfg.tracker` (MaskHungarianAssignerVideo + MaskPseudoSampler): losses, the per-stage clip assignments, gradients w.r.t.
    x / the per-frame object features and a sample of the parameters."""
    import knet_vis.det.mask_hungarian_assigner  # noqa: F401  (registers DiceCost / MaskCost)
    import knet_vis.tracker.mask_hungarian_assigner  # noqa: F401  (registers MaskHungarianAssignerVideo)
    import knet_vis.det.mask_pseudo_sampler  # noqa: F401
    train_cfg = AttrDict(assigner=dict(type='MaskHungarianAssignerVideo', cls_cost=dict(type='FocalLossCost', weight=2.0),
                                       dice_cost=dict(type='DiceCost', weight=4.0, pred_act=True),
                                       mask_cost=dict(type='MaskCost', weight=1.0, pred_act=True)),
                         sampler=dict(type='MaskPseudoSampler'), pos_weight=1)
    trk = build_head(dict(type='KernelFrameIterHeadVideo', num_proposals=N, num_stages=3, assign_stages=2, proposal_feature_channel=C,
                          stage_loss_weights=(1., 1., 1.), num_thing_classes=ncls, num_stuff_classes=0, train_cfg=train_cfg,
                          query_merge_method=merge, with_mask_init=mask_init,
                          mask_head=stage_cfg('KernelUpdateHeadVideo', C, heads, ffn, ncls, up, num_proposals=N, query_merge_method=merge)))
    trk.train()
    shapes = {k: tuple(v.shape) for k, v in trk.state_dict().items()}
    trk.load_state_dict({k: torch.from_numpy(v) for k, v in synth.state_dict_like(shapes, seed + 1).items()}, strict=True)
    B = bs * nf
    x, pf, mp = (torch.from_numpy(a) for a in synth.head_inputs(B, N, C, H, W, seed))
    x = x.reshape(bs, nf, C, H, W).requires_grad_(True)
    obj = pf.reshape(bs, nf, N, C, 1, 1).requires_grad_(True)
    masks = mp.reshape(bs, nf, N, H, W)
    tg = synth.clip_targets(bs, nf, ncls, H * up, W * up, seed)
    gt_masks = [[torch.from_numpy(m) for m in t['gt_masks']] for t in tg]
    gt_labels = [torch.from_numpy(t['gt_labels']) for t in tg]
    gt_ids = [torch.from_numpy(t['gt_instance_ids']) for t in tg]
    ref_img_metas = [[dict()] * nf for _ in range(bs)]
    assigned = []
    for a in trk.mask_assigner:
        orig = a.assign

        def rec(*args, _orig=orig, **kw):
            r = _orig(*args, **kw)
            assigned.append(r[0].gt_inds.clone())
            return r
        a.assign = rec
    losses, feats = trk.forward_train(x, ref_img_metas, None, masks, obj, gt_masks, gt_labels, gt_ids)
    total = sum(v for k, v in losses.items() if 'loss' in k) + 0.01 * (feats['obj_feats'] ** 2).sum()
    total.backward()
    out = dict(case=np.array([C, heads, ffn, ncls, N, H, W, up, bs, nf, seed, int(mask_init)], dtype=np.int64), merge=np.array(merge),
               keys=np.array(sorted(shapes)), shapes=np.array([str(shapes[k]) for k in sorted(shapes)]),
               loss_keys=np.array(sorted(losses)), loss_vals=np.array([float(losses[k]) for k in sorted(losses)], dtype=np.float64),
               total=np.float64(float(total)), assigned=torch.stack(assigned).numpy(), feat_obj=feats['obj_feats'].detach().numpy(),
               feat_masks=feats['masks'].detach().numpy(), grad_x=x.grad.numpy(), grad_obj=obj.grad.numpy())
    named = dict(trk.named_parameters())
    gk = [k for k in TRAIN_GRAD_KEYS if k in named]
    out['grad_keys'] = np.array(gk)
    for i, k in enumerate(gk):
        out[f'grad_{i}'] = named[k].grad.numpy()
    out['all_keys'] = np.array(sorted(named))
    out['all_gnorm'] = np.array([float(named[k].grad.double().norm()) if named[k].grad is not None else -1.0 for k in sorted(named)])
    np.savez_compressed(os.path.join(OUT, name + '.npz'), **out)
    print(f'{name}: ok  total={float(total):.5f}  ' + ' '.join(f'{k}={float(v):.4f}' for k, v in sorted(losses.items())[:6]))


def run_rpn_roi_train(name, C, heads, ffn, ncls, nprop, H, W, up, S, bs, nf, seed):
    """The VIS model's rpn_head -> roi_head training hand-over on bs clips of nf frames: ConvKernelHeadVideo.forward_train behind a
    pass-through neck (things only, sigmoid focal semantic loss as in configs/video_knet_vis/.../knet_track_r50_1x_youtubevis.py),
    then KernelIterHeadVideo.forward_train on its outputs.  Stored: both loss dicts, every Hungarian assignment, the roi head's
    clip-shaped features."""
    from mmdet.models.builder import NECKS

    if 'PassThroughNeckVis' not in NECKS._module_dict:
        @NECKS.register_module()
        class PassThroughNeckVis(torch.nn.Module):
            def forward(self, feats):
                return [feats[0], feats[1]] if len(feats) == 2 else feats[0]
    assign = dict(type='MaskHungarianAssigner', cls_cost=dict(type='FocalLossCost', weight=2.0),
                  dice_cost=dict(type='DiceCost', weight=4.0, pred_act=True), mask_cost=dict(type='MaskCost', weight=1.0, pred_act=True))
    rpn = build_head(dict(type='ConvKernelHeadVideo', num_proposals=nprop, in_channels=C, out_channels=C, num_loc_convs=0,
                          num_seg_convs=0, localization_fpn=dict(type='PassThroughNeckVis'), conv_kernel_size=1, semantic_fpn=True,
                          num_classes=ncls, use_binary=True, proposal_feats_with_obj=True, feat_downsample_stride=up, feat_refine=False,
                          num_thing_classes=ncls, num_stuff_classes=0, cat_stuff_mask=False,
                          loss_seg=dict(type='FocalLoss', use_sigmoid=True, gamma=2.0, alpha=0.25, loss_weight=1.0),
                          loss_mask=dict(type='CrossEntropyLoss', use_sigmoid=True, loss_weight=1.0),
                          loss_dice=dict(type='DiceLoss', loss_weight=4.0),
                          train_cfg=AttrDict(assigner=assign, sampler=dict(type='MaskPseudoSampler'), pos_weight=1)))
    roi = build_head(dict(type='KernelIterHeadVideo', num_stages=S, stage_loss_weights=[1] * S, assign_stages=S, proposal_feature_channel=C,
                          num_thing_classes=ncls, num_stuff_classes=0, num_proposals=nprop,
                          train_cfg=[AttrDict(assigner=assign, sampler=dict(type='MaskPseudoSampler'), pos_weight=1) for _ in range(S)],
                          mask_head=[stage_cfg('KernelUpdateHead', C, heads, ffn, ncls, up) for _ in range(S)]))
    for m, sd_seed in ((rpn, seed), (roi, seed + 1)):
        shapes = {k: tuple(v.shape) for k, v in m.state_dict().items()}
        m.load_state_dict({k: torch.from_numpy(v) for k, v in synth.state_dict_like(shapes, sd_seed).items()}, strict=True)
        m.train()
    F = bs * nf
    loc = torch.from_numpy(synth.uniform((F, C, H, W), seed + 2, -1.0, 1.0)).requires_grad_(True)
    sem = torch.from_numpy(synth.uniform((F, C, H, W), seed + 3, -1.0, 1.0)).requires_grad_(True)
    tg = synth.clip_targets(bs, nf, ncls, H * up, W * up, seed)
    gt_masks = [[torch.from_numpy(m) for m in t['gt_masks']] for t in tg]
    gt_labels = [torch.from_numpy(t['gt_labels']) for t in tg]
    metas = [[dict() for _ in range(nf)] for _ in range(bs)]
    assigned = []

    def hook(a):
        orig = a.assign

        def rec(*args, **kw):
            r = orig(*args, **kw)
            assigned.append(r.gt_inds.clone())
            return r
        a.assign = rec
    hook(rpn.assigner)
    for a in roi.mask_assigner:
        hook(a)
    rl, prop, x_feats, masks, cls = rpn.forward_train((loc, sem), [dict()] * bs, metas, gt_masks, gt_labels)
    n_rpn = len(assigned)
    ll, feats = roi.forward_train(x_feats, prop, masks, cls, metas, gt_masks, gt_labels)
    total = sum(v for k, v in rl.items() if 'loss' in k) + sum(v for k, v in ll.items() if 'loss' in k)
    total.backward()
    out = dict(case=np.array([C, heads, ffn, ncls, nprop, H, W, up, S, bs, nf, seed], dtype=np.int64),
               rpn_keys=np.array(sorted(rl)), rpn_vals=np.array([float(rl[k].detach()) for k in sorted(rl)], dtype=np.float64),
               roi_keys=np.array(sorted(ll)), roi_vals=np.array([float(ll[k].detach()) for k in sorted(ll)], dtype=np.float64),
               assigned_rpn=torch.stack(assigned[:n_rpn]).numpy(), assigned_roi=torch.stack(assigned[n_rpn:]).numpy(),
               proposal_feats=prop.detach().numpy(), feat_obj=feats['obj_feats'].detach().numpy(),
               feat_cls=feats['cls_scores'].detach().numpy(), feat_mask_rowsum=feats['masks'].detach().double().sum(dim=(-1, -2)).numpy(),
               grad_loc_norm=np.float64(float(loc.grad.double().norm())), grad_sem_norm=np.float64(float(sem.grad.double().norm())),
               rpn_keys_sd=np.array(sorted(rpn.state_dict())), total=np.float64(float(total.detach())))
    np.savez_compressed(os.path.join(OUT, name + '.npz'), **out)
    print(f'{name}: ok  total={float(total):.5f}  rpn ' + ' '.join(f'{k}={float(v):.4f}' for k, v in sorted(rl.items())))


if __name__ == '__main__':
    torch.manual_seed(0)
    torch.set_num_threads(8)
    run('vis_tiny', C=64, heads=8, ffn=128, ncls=7, N=20, H=8, W=16, up=2, S=2, bs=2, nf=3, seed=91, kmax=10)
    run('vis_cfg', C=256, heads=8, ffn=2048, ncls=40, N=100, H=12, W=20, up=2, S=3, bs=1, nf=2, seed=93, kmax=10)
    # query_merge_method 'attention' / 'attention_pos' (no shipped config sets them; the classes build and run them)
    run('vis_attn_tiny', C=64, heads=8, ffn=128, ncls=7, N=20, H=8, W=16, up=2, S=2, bs=2, nf=3, seed=95, kmax=10, merge='attention')
    run('vis_attnpos_tiny', C=64, heads=8, ffn=128, ncls=7, N=20, H=8, W=16, up=2, S=2, bs=2, nf=3, seed=96, kmax=10, merge='attention_pos')
    run('vis_attnpos_cfg', C=256, heads=8, ffn=2048, ncls=40, N=100, H=12, W=20, up=2, S=3, bs=1, nf=3, seed=97, kmax=10, merge='attention_pos')
    # clip-level training of the tracker head (MaskHungarianAssignerVideo)
    run_train('vis_train_tiny', C=64, heads=8, ffn=128, ncls=7, N=20, H=8, W=16, up=2, bs=2, nf=3, seed=101)
    run_train('vis_train_attnpos', C=64, heads=8, ffn=128, ncls=7, N=20, H=8, W=16, up=2, bs=2, nf=3, seed=102, merge='attention_pos',
              mask_init=True)
    run_rpn_roi_train('vis_rpn_train', C=64, heads=8, ffn=128, ncls=7, nprop=20, H=8, W=16, up=2, S=2, bs=2, nf=3, seed=111)

#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the REFERENCE ITSELF — build-container only (needs /root/reference).

The reference's own hot-path files are imported UNMODIFIED from /root/reference (read-only; bytecode writing
disabled) through the plumbing stand-ins in oracle/standins/ (mmcv/mmdet are not installable here, see
oracle/standins/README.md).  Modules are built with `build_head(cfg)` exactly as the reference's detector does,
loaded with hash-formula weights (oracle/synth.py) and run on hash-formula inputs; ONLY the reference's outputs
(plus the case parameters needed to regenerate weights/inputs) are stored.  No reference source text is stored.

    python oracle/gen_golden.py            # rewrites tests/golden/*.npz

Cases:
  det_tiny      C=64  N=12+3  8x16   S=3 B=2  x2   every per-stage intermediate, full tensors
  det_odd       C=64  N=21    9x15   S=2 B=1  x2   ragged P (=135), N not a multiple of anything
  det_cfg       C=256 N=117   16x32  S=3 B=2  x2   config-size head, full outputs
  det_cfg_big   C=256 N=117   64x128 S=3 B=1  x2   BASELINE cfg1 size: sampled logits + row sums + packed sign bits
  video_tiny / video_cfg      VideoKernelIterHead, previous_type='ffn', x4 upsample, 5th (tracking) output
  thr_kat       sigmoid(z) > 0.5 around the fp32 flip point (z in {0, +-5e-8, 8.9e-8, +-1e-7, ...})
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

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

from oracle import synth  # noqa: E402

import knet.kernel_updator  # noqa: E402,F401  (registers KernelUpdator)
import knet.det.kernel_update_head  # noqa: E402,F401
import knet.det.kernel_iter_head  # noqa: E402,F401
import knet.video.kernel_update_head  # noqa: E402,F401
import knet.video.kernel_iter_head  # noqa: E402,F401
import knet.det.kernel_head  # noqa: E402,F401  (ConvKernelHead: the kernel-initialisation pass)
import knet.det.mask_hungarian_assigner  # noqa: E402,F401  (MaskHungarianAssigner, DiceCost, MaskCost)
import knet.det.mask_pseudo_sampler  # noqa: E402,F401  (MaskPseudoSampler)
import knet.cross_entropy_loss  # noqa: E402,F401  (the reference's own CrossEntropyLoss, registered with force=True)
from mmdet.models.builder import NECKS, build_head  # noqa: E402

OUT = os.environ.get('VKN_GOLDEN_OUT', os.path.join(ROOT, 'tests', 'golden'))   # tests/test_golden_regen.py regenerates into a tmp dir


def head_cfg(video, C, heads, ffn, ncls, n_thing, n_stuff, S, up, nprop, plink=None, ptype='ffn'):
    """Same dict layout as configs/det/_base_/models/knet_kitti_step_s3_r50_fpn.py:79-138 (det) and
    configs/det/video_knet_kitti_step/video_knet_s3_r50_*_link_ffn_joint_train.py:78-137 (video)."""
    mh = dict(
        type='VideoKernelUpdateHead' if video else 'KernelUpdateHead',
        num_classes=ncls, num_thing_classes=n_thing, num_stuff_classes=n_stuff, num_ffn_fcs=2, num_heads=heads,
        num_cls_fcs=1, num_mask_fcs=1, feedforward_channels=ffn, in_channels=C, out_channels=C, dropout=0.0,
        mask_thr=0.5, conv_kernel_size=1, mask_upsample_stride=up, ffn_act_cfg=dict(type='ReLU', inplace=True),
        with_ffn=True, feat_transform_cfg=dict(conv_cfg=dict(type='Conv2d'), act_cfg=None),
        kernel_updator_cfg=dict(type='KernelUpdator', in_channels=C, feat_channels=C, out_channels=C,
                                input_feat_shape=3, act_cfg=dict(type='ReLU', inplace=True), norm_cfg=dict(type='LN')),
        loss_rank=dict(type='CrossEntropyLoss', use_sigmoid=False, loss_weight=0.1),
        loss_mask=dict(type='CrossEntropyLoss', use_sigmoid=True, loss_weight=1.0),
        loss_dice=dict(type='DiceLoss', loss_weight=4.0),
        loss_cls=dict(type='FocalLoss', use_sigmoid=True, gamma=2.0, alpha=0.25, loss_weight=2.0))
    if video:
        # previous_link / previous_type: configs/det/video_knet_kitti_step/video_knet_s3_swin{b,l}_*_joint_update.py:98-100
        # ('update_dynamic_cov' + 'update'), ..._update_conv_short_track_fc.py:95-97 ('update_dynamic_cov' + 'ffn')
        mh.update(previous='placeholder', previous_type=ptype)
        if plink is not None:
            mh.update(previous_link=plink)
    import copy
    cfg = dict(type='VideoKernelIterHead' if video else 'KernelIterHead', num_thing_classes=n_thing,
               num_stuff_classes=n_stuff, num_stages=S, stage_loss_weights=[1] * S, proposal_feature_channel=C,
               num_proposals=nprop, mask_head=[copy.deepcopy(mh) for _ in range(S)])
    if video:
        cfg.update(with_track=True, merge_joint=True)
    else:
        cfg.update(do_panoptic=True)
    return cfg


def build_reference(video, **kw):
    head = build_head(head_cfg(video, **kw))
    head.eval()
    shapes = {k: tuple(v.shape) for k, v in head.state_dict().items()}
    return head, shapes


def load_formula_weights(head, shapes, seed):
    sd = {k: torch.from_numpy(v) for k, v in synth.state_dict_like(shapes, seed).items()}
    head.load_state_dict(sd, strict=True)
    return sd


def pack_signs(t, margin=2e-3):
    """bits of (t > 0) and a validity mask |t| > margin, both packed."""
    a = t.detach().numpy().ravel()
    return np.packbits(a > 0), np.packbits(np.abs(a) > margin)


CASES = {
    'det_tiny': dict(video=False, C=64, heads=8, ffn=128, ncls=5, n_thing=2, n_stuff=3, S=3, up=2, nprop=12, N=15, H=8, W=16, B=2, seed=1),
    'det_odd': dict(video=False, C=64, heads=8, ffn=128, ncls=7, n_thing=7, n_stuff=0, S=2, up=2, nprop=21, N=21, H=9, W=15, B=1, seed=2),
    'det_cfg': dict(video=False, C=256, heads=8, ffn=2048, ncls=19, n_thing=2, n_stuff=17, S=3, up=2, nprop=100, N=117, H=16, W=32, B=2, seed=3),
    'det_cfg_big': dict(video=False, C=256, heads=8, ffn=2048, ncls=19, n_thing=2, n_stuff=17, S=3, up=2, nprop=100, N=117, H=64, W=128, B=1, seed=4),
    'video_tiny': dict(video=True, C=64, heads=8, ffn=128, ncls=5, n_thing=2, n_stuff=3, S=3, up=4, nprop=12, N=15, H=8, W=16, B=2, seed=5),
    'video_cfg': dict(video=True, C=256, heads=8, ffn=2048, ncls=19, n_thing=2, n_stuff=17, S=3, up=4, nprop=100, N=117, H=16, W=32, B=1, seed=6),
    # BASELINE cfg5 at its real size: video_knet_s3_swinb VIP-Seg, 720p -> 92x160 stride-8 features, 100 + 66 kernels, 124 classes, x4
    'video_vipseg_big': dict(video=True, C=256, heads=8, ffn=2048, ncls=124, n_thing=58, n_stuff=66, S=3, up=4, nprop=100, N=166, H=92, W=160,
                             B=1, seed=8),
    # BASELINE cfg4 per-frame shape (YouTube-VIS 640x360 -> 48x80): N = 100, 40 thing classes, no stuff, x2 — through the knet head
    # (the knet_vis copy of the stage is the same arithmetic; its registry names are pinned by oracle/gen_golden_vis.py)
    # the "update" video heads: the LAST stage's kernels are rewritten from the previous frame's kernels before the update
    # (previous_link='update_dynamic_cov', knet/video/kernel_update_head.py:324-348), tracking embedding through a second
    # KernelUpdator (previous_type='update', :417-445) or the plain attention link (previous_type='ffn')
    'video_upd_tiny': dict(video=True, C=64, heads=8, ffn=128, ncls=5, n_thing=2, n_stuff=3, S=3, up=4, nprop=12, N=15, H=8, W=16, B=3, seed=11,
                           plink='update_dynamic_cov', ptype='update'),
    'video_updffn_tiny': dict(video=True, C=64, heads=8, ffn=128, ncls=5, n_thing=2, n_stuff=3, S=2, up=2, nprop=12, N=15, H=8, W=16, B=2,
                              seed=12, plink='update_dynamic_cov', ptype='ffn'),
    'video_upd_cfg': dict(video=True, C=256, heads=8, ffn=2048, ncls=19, n_thing=2, n_stuff=17, S=3, up=4, nprop=100, N=117, H=16, W=32, B=2,
                          seed=13, plink='update_dynamic_cov', ptype='update'),
    # the link combinations no shipped config uses (the ctor accepts them): attention-only previous link (`link_atten`, :350-372) with the
    # updator tracking link, and the `update_obj` tracking link (:446-476) without / with a previous link
    'video_latt_upd_tiny': dict(video=True, C=64, heads=8, ffn=128, ncls=5, n_thing=2, n_stuff=3, S=2, up=2, nprop=12, N=15, H=8, W=16, B=3,
                                seed=15, plink='link_atten', ptype='update'),
    'video_updobj_tiny': dict(video=True, C=64, heads=8, ffn=128, ncls=5, n_thing=2, n_stuff=3, S=2, up=2, nprop=12, N=15, H=8, W=16, B=3,
                              seed=16, plink=None, ptype='update_obj'),
    'video_latt_updobj_tiny': dict(video=True, C=64, heads=8, ffn=128, ncls=5, n_thing=2, n_stuff=3, S=2, up=2, nprop=12, N=15, H=8, W=16,
                                   B=2, seed=17, plink='link_atten', ptype='update_obj'),
    # BASELINE cfg5 as literally worded: 150 proposals + 66 stuff kernels = 216 rows (the reference's VIP-Seg config has 100 + 66)
    'video_vipseg_n216': dict(video=True, C=256, heads=8, ffn=2048, ncls=124, n_thing=58, n_stuff=66, S=3, up=4, nprop=150, N=216, H=92, W=160,
                              B=1, seed=14),
    'det_ytvis': dict(video=False, C=256, heads=8, ffn=2048, ncls=40, n_thing=40, n_stuff=0, S=3, up=2, nprop=100, N=100, H=48, W=80, B=2,
                      seed=9),
}


def run_case(name, p):
    p = dict(p)
    N, H, W, B, seed = (p.pop(k) for k in ('N', 'H', 'W', 'B', 'seed'))
    video = p['video']
    head, shapes = build_reference(**p)
    load_formula_weights(head, shapes, seed)
    x, pf, mp = (torch.from_numpy(a) for a in synth.head_inputs(B, N, p['C'], H, W, seed))
    metas = [dict() for _ in range(B)]
    out = dict(case=np.array([p['C'], p['heads'], p['ffn'], p['ncls'], p['n_thing'], p['n_stuff'], p['S'], p['up'],
                              p['nprop'], N, H, W, B, seed, int(video)], dtype=np.int64),
               keys=np.array(sorted(shapes)), shapes=np.array([str(shapes[k]) for k in sorted(shapes)]))
    if p.get('plink') is not None or p.get('ptype', 'ffn') != 'ffn':
        out['plink'], out['ptype'] = np.array(p['plink'] or ''), np.array(p['ptype'])
    with torch.no_grad():
        # per-stage intermediates straight from the reference's stage modules
        obj, masks = pf, mp
        per_stage = []
        for s in range(p['S']):
            r = head.mask_head[s](x, obj, masks, img_metas=metas)
            cls, masks, obj = r[0], r[1], r[2]
            per_stage.append((cls, masks, obj))
        if video:
            prev = torch.from_numpy(synth.normalish((B, N, p['C'], 1, 1), 99 + seed, 1.0))
            o, c, m, sc = head.simple_test_mask_preds_plus_previous(x, pf, mp, None, metas, previous_obj_feats=prev)
            # the harness drops the tracking output; take it from the last stage directly (video kernel_iter_head.py:118-148)
            obj2, masks2 = pf, mp
            for s in range(p['S']):
                mr = head._mask_forward(s, x, obj2, masks2, metas, previous_obj_feats=prev if s == p['S'] - 1 else None)
                obj2, masks2 = mr['object_feats'], mr['mask_preds']
            out['track'] = mr['object_feats_track'].numpy()
            assert torch.equal(masks2, m)
            if p.get('plink') is not None or p.get('ptype', 'ffn') != 'ffn':
                # the B frames as CONSECUTIVE frames of one video, walked the way the detector does
                # (knet/video/knet_quansi_dense_embed_fc_joint_train.py:505-525): frame 0 has no previous kernels, frame t > 0 gets
                # frame t - 1's last-stage object_feats — with previous_link the masks of frame t depend on frame t - 1
                memo = None
                for t in range(B):
                    obj3, masks3 = pf[t:t + 1], mp[t:t + 1]
                    for s in range(p['S']):
                        mr3 = head._mask_forward(s, x[t:t + 1], obj3, masks3, metas[:1],
                                                 previous_obj_feats=memo if s == p['S'] - 1 else None)
                        obj3, masks3 = mr3['object_feats'], mr3['mask_preds']
                    memo = obj3
                    out[f'clip_obj{t}'] = obj3.numpy()
                    out[f'clip_cls{t}'] = mr3['cls_score'].sigmoid().numpy()
                    if p['C'] <= 64:
                        out[f'clip_mask{t}'] = masks3.numpy()
                    else:
                        out[f'clip_mask_rowsum{t}'] = masks3.double().sum(dim=(-1, -2)).numpy()
                    if mr3['object_feats_track'] is not None:
                        out[f'clip_track{t}'] = mr3['object_feats_track'].numpy()
        else:
            o, c, m, sc = head.simple_test_mask_preds(x, pf, mp, None, metas)
    assert p.get('plink') is not None or torch.equal(per_stage[-1][1], m)   # (previous_link rewrites the last stage's kernels)
    # per (stage, frame, kernel): the smallest |logit - flip point| of the mask that stage hands to the next gather — a kernel whose
    # margins all exceed the fp32 noise of a different summation order cannot flip a bit (free-running parity at large sizes)
    flip = 8.940697e-08
    out['row_margin'] = np.stack([(ms - flip).abs().flatten(2).min(dim=2).values.numpy() for _, ms, _ in per_stage[:-1]])
    big = name.endswith('_big') or name in ('det_ytvis', 'video_vipseg_n216')
    out['object_feats'] = o.numpy()
    out['cls_score'] = c.numpy()
    if not big:
        out['mask_preds'] = m.numpy()
        for s, (cls_s, m_s, o_s) in enumerate(per_stage):
            out[f's{s}_cls'] = cls_s.numpy()
            out[f's{s}_obj'] = o_s.numpy()
            if p['C'] <= 64:
                out[f's{s}_mask'] = m_s.numpy()
        if p['C'] <= 64:
            out['scaled_mask_preds'] = sc.numpy()
        else:
            out['scaled_rowsum'] = sc.double().sum(dim=(-1, -2)).numpy()
    else:
        flat = m.reshape(-1)
        idx = (synth.uniform((4096,), 4242, 0.0, 1.0).astype(np.float64) * flat.numel()).astype(np.int64)
        out['sample_idx'] = idx
        out['sample_val'] = flat[idx].numpy()
        out['mask_rowsum'] = m.double().sum(dim=(-1, -2)).numpy()
        out['mask_rowabs'] = m.double().abs().sum(dim=(-1, -2)).numpy()
        out['scaled_rowsum'] = sc.double().sum(dim=(-1, -2)).numpy()
        out['sign_bits'], out['sign_valid'] = pack_signs(m)
    np.savez_compressed(os.path.join(OUT, name + '.npz'), **out)
    print(f'{name}: ok  max|mask|={float(m.abs().max()):.2f}')


@NECKS.register_module()
class PassThroughNeck(torch.nn.Module):
    """Test plumbing: stands where `SemanticFPNWrapper` sits and hands the two feature maps straight through, so that
    `ConvKernelHead._decode_init_proposals` runs exactly the part SURVEY.md §8(f)-2 scopes (no loc/seg convs)."""

    def forward(self, feats):
        return [feats[0], feats[1]] if len(feats) == 2 else feats[0]


INIT_CASES = {
    'init_tiny': dict(C=64, nprop=12, ncls=5, n_thing=2, H=8, W=16, B=2, seed=21, sem=True, cat=True),
    'init_odd': dict(C=64, nprop=21, ncls=0, n_thing=0, H=9, W=15, B=1, seed=22, sem=False, cat=False),
    'init_cfg': dict(C=256, nprop=100, ncls=19, n_thing=2, H=16, W=32, B=2, seed=23, sem=True, cat=True),
    # use_binary=False: gather weights (sigmoid(z) > 0.5) * sigmoid(z)  (knet/det/kernel_head.py:246-247)
    'init_soft': dict(C=64, nprop=12, ncls=5, n_thing=2, H=8, W=16, B=2, seed=24, sem=True, cat=True, soft=True),
}


def init_inputs(p):
    """loc / semantic features and the ConvKernelHead weights of an init case (hash-formula, regenerated by the tests)."""
    B, C, H, W, seed = p['B'], p['C'], p['H'], p['W'], p['seed']
    loc = synth.normalish((B, C, H, W), 31 + 7 * seed, 1.0)
    sem = synth.normalish((B, C, H, W), 32 + 7 * seed, 1.0) if p['sem'] else None
    shapes = {'init_kernels.weight': (p['nprop'], C, 1, 1)}
    if p['sem']:
        shapes['conv_seg.weight'] = (p['ncls'], C, 1, 1)
        shapes['conv_seg.bias'] = (p['ncls'],)
    return loc, sem, shapes


def run_init_case(name, p):
    """ConvKernelHead.simple_test_rpn of the reference (knet/det/kernel_head.py:506-508) behind a pass-through neck."""
    cfg = dict(type='ConvKernelHead', num_proposals=p['nprop'], in_channels=p['C'], out_channels=p['C'], num_loc_convs=0,
               num_seg_convs=0, localization_fpn=dict(type='PassThroughNeck'), conv_kernel_size=1, semantic_fpn=p['sem'],
               num_classes=max(p['ncls'], 1), use_binary=not p.get('soft', False), proposal_feats_with_obj=True, feat_downsample_stride=1,
               num_thing_classes=p['n_thing'], num_stuff_classes=p['ncls'] - p['n_thing'], cat_stuff_mask=p['cat'])
    head = build_head(cfg)
    head.eval()
    loc, sem, shapes = init_inputs(p)
    assert {k: tuple(v.shape) for k, v in head.state_dict().items()} == shapes, head.state_dict().keys()
    sd = {k: torch.from_numpy(v) for k, v in synth.state_dict_like(shapes, p['seed']).items()}
    head.load_state_dict(sd, strict=True)
    feats = (torch.from_numpy(loc), torch.from_numpy(sem)) if p['sem'] else (torch.from_numpy(loc),)
    with torch.no_grad():
        prop, x_feats, masks, cls, seg = head.simple_test_rpn(feats, [dict() for _ in range(p['B'])])
    assert cls is None
    out = dict(case=np.array([p['C'], p['nprop'], p['ncls'], p['n_thing'], p['H'], p['W'], p['B'], p['seed'], int(p['sem']),
                              int(p['cat'])], dtype=np.int64),
               keys=np.array(sorted(shapes)), shapes=np.array([str(shapes[k]) for k in sorted(shapes)]),
               proposal_feats=prop.numpy(), mask_preds=masks.numpy())
    if p['C'] <= 64:
        out['x_feats'] = x_feats.numpy()
    else:
        out['x_feats_rowsum'] = x_feats.double().sum(dim=(-1, -2)).numpy()
    if seg is not None:
        out['seg_preds'] = seg.numpy()
    np.savez_compressed(os.path.join(OUT, name + '.npz'), **out)
    frac = float((masks[:, :p['nprop']] > 0).float().mean())
    print(f'{name}: ok  N={masks.shape[1]} on-fraction={frac:.3f} max|prop|={float(prop.abs().max()):.1f}')


def init_keys():
    """state-dict keys / shapes of the reference's ConvKernelHead under the shipped rpn_head kwargs
    (configs/det/_base_/models/knet_kitti_step_s3_r50_fpn.py:29-61; neck replaced by the pass-through), plus the
    feat_refine=True variant that adds ins_downsample / seg_downsample."""
    out = {}
    for tag, refine in (('kitti', False), ('refine', True)):
        cfg = dict(type='ConvKernelHead', num_classes=19, num_thing_classes=2, num_stuff_classes=17, cat_stuff_mask=True,
                   conv_kernel_size=1, feat_downsample_stride=2, feat_refine_stride=1, feat_refine=refine, use_binary=True,
                   num_loc_convs=1, num_seg_convs=1, conv_normal_init=True, localization_fpn=dict(type='PassThroughNeck'),
                   num_proposals=100, proposal_feats_with_obj=True, xavier_init_kernel=False, kernel_init_std=1)
        sd = build_head(cfg).state_dict()
        out[tag + '_keys'] = np.array(sorted(sd))
        out[tag + '_shapes'] = np.array([str(tuple(sd[k].shape)) for k in sorted(sd)])
    np.savez_compressed(os.path.join(OUT, 'init_keys.npz'), **out)
    print('init_keys: ok', len(out['kitti_keys']), len(out['refine_keys']))


class AttrDict(dict):
    """mmcv.Config-style attribute access for test_cfg (the reference reads `test_cfg.max_per_img`, `merge_cfg.overlap_thr`)."""
    __getattr__ = dict.__getitem__


PAN_CASES = {
    # three resampling levels: x2, -> batch input, crop, -> ori (up-scaling by 1.5)
    'pan_tiny': dict(B=2, N=15, Np=12, T=2, ncls=5, Hm=8, Wm=16, up=2, bis=(64, 128), img=(60, 120), ori=(90, 180), seed=31),
    # last resize is the identity (Cityscapes-style: ori == img == batch input)
    'pan_ident': dict(B=2, N=15, Np=12, T=2, ncls=5, Hm=8, Wm=16, up=4, bis=(64, 128), img=(64, 128), ori=(64, 128), seed=32),
    # config kernel count, down-scaling last level, odd crop
    'pan_cfg': dict(B=1, N=117, Np=100, T=2, ncls=19, Hm=32, Wm=64, up=4, bis=(256, 512), img=(250, 499), ori=(125, 250), seed=33),
    # VIP-Seg class layout: 58 thing classes (5800 (proposal, class) candidates for the top-k), 66 stuff kernels
    'pan_vipseg': dict(B=1, N=166, Np=100, T=58, ncls=124, Hm=24, Wm=40, up=4, bis=(192, 320), img=(184, 320), ori=(184, 320), seed=35),
    # KITTI-like odd sizes, already-scaled logits (up = 1), crop only
    'pan_kitti': dict(B=1, N=117, Np=100, T=2, ncls=19, Hm=48, Wm=156, up=1, bis=(96, 312), img=(94, 311), ori=(94, 311), seed=34),
}


def run_pan_video_case(name, p):
    """VideoKernelIterHead.get_panoptic of the reference (knet/video/kernel_iter_head.py:591-640, merge_stuff_thing_stuff_joint
    :832-905): the 5-tuple incl. `thing_obj_feat = sort_obj_fea[things_ids]` — the tracking embeddings of the accepted things."""
    test_cfg = AttrDict(max_per_img=p['Np'], mask_thr=0.5, stuff_score_thr=0.05,
                        merge_stuff_thing=AttrDict(overlap_thr=0.6, iou_thr=0.5, stuff_max_area=4096, instance_score_thr=0.25))
    C = 32
    cfg = head_cfg(True, C=C, heads=8, ffn=64, ncls=p['ncls'], n_thing=p['T'], n_stuff=p['ncls'] - p['T'], S=1, up=p['up'], nprop=p['Np'])
    cfg.update(with_track=True, merge_joint=True, test_cfg=test_cfg)
    head = build_head(cfg)
    head.eval()
    cls, logits = (torch.from_numpy(a) for a in synth.panoptic_inputs(p['B'], p['N'], p['Np'], p['ncls'], p['Hm'], p['Wm'], p['seed']))
    obj = torch.from_numpy(synth.normalish((p['B'], p['N'], C), 77 + p['seed'], 1.0))
    meta = dict(img_shape=(*p['img'], 3), batch_input_shape=tuple(p['bis']), ori_shape=(*p['ori'], 3))
    out = dict(case=np.array([p['B'], p['N'], p['Np'], p['T'], p['ncls'], p['Hm'], p['Wm'], p['up'], *p['bis'], *p['img'], *p['ori'],
                              p['seed']], dtype=np.int64))
    with torch.no_grad():
        scaled = F.interpolate(logits, scale_factor=p['up'], align_corners=False, mode='bilinear') if p['up'] > 1 else logits
        segs, nseg = [], []
        for b in range(p['B']):
            bboxes, _, _, (pan, info), tfeat = head.get_panoptic(cls[b], scaled[b], head.test_cfg, meta, obj_feat=obj[b])
            segs.append(pan)
            nseg.append(len(info))
            out[f'info{b}'] = np.array([[s_['id'], int(s_['isthing']), s_['category_id'], s_.get('instance_id', -1),
                                         s_.get('score', float('nan')), s_.get('area', -1)] for s_ in info], dtype=np.float64).reshape(-1, 6)
            out[f'thing_obj_feat{b}'] = tfeat.numpy()
    out['panoptic_seg'] = np.stack(segs).astype(np.int32)
    out['nseg'] = np.array(nseg, dtype=np.int64)
    np.savez_compressed(os.path.join(OUT, name + '.npz'), **out)
    print(f'{name}: ok  segments per frame = {nseg}  things with embeddings = {[out[f"thing_obj_feat{b}"].shape[0] for b in range(p["B"])]}')


def run_pan_thing_first(name, p):
    """KernelIterHead.get_panoptic with merge_joint=False -> merge_stuff_thing (knet/det/kernel_iter_head.py:332-370, 385-465)."""
    test_cfg = AttrDict(max_per_img=p['Np'], mask_thr=0.5, stuff_score_thr=0.05,
                        merge_stuff_thing=AttrDict(overlap_thr=0.6, iou_thr=0.5, stuff_max_area=p.get('sma', 64), instance_score_thr=0.25))
    cfg = head_cfg(False, C=32, heads=8, ffn=64, ncls=p['ncls'], n_thing=p['T'], n_stuff=p['ncls'] - p['T'], S=1, up=p['up'], nprop=p['Np'])
    cfg.update(do_panoptic=True, merge_joint=False, test_cfg=test_cfg)
    head = build_head(cfg)
    head.eval()
    cls, logits = (torch.from_numpy(a) for a in synth.panoptic_inputs(p['B'], p['N'], p['Np'], p['ncls'], p['Hm'], p['Wm'], p['seed']))
    meta = dict(img_shape=(*p['img'], 3), batch_input_shape=tuple(p['bis']), ori_shape=(*p['ori'], 3))
    out = dict(case=np.array([p['B'], p['N'], p['Np'], p['T'], p['ncls'], p['Hm'], p['Wm'], p['up'], *p['bis'], *p['img'], *p['ori'],
                              p['seed']], dtype=np.int64), stuff_max_area=np.int64(p.get('sma', 64)))
    segs, nseg = [], []
    with torch.no_grad():
        scaled = F.interpolate(logits, scale_factor=p['up'], align_corners=False, mode='bilinear') if p['up'] > 1 else logits
        for b in range(p['B']):
            bbox_result, segm_result, (pan, info) = head.get_panoptic(cls[b], scaled[b], head.test_cfg, meta)
            segs.append(pan)
            nseg.append(len(info))
            out[f'info{b}'] = np.array([[s_['id'], int(s_['isthing']), s_['category_id'], s_.get('instance_id', -1),
                                         s_.get('score', float('nan')), s_.get('area', -1)] for s_ in info], dtype=np.float64).reshape(-1, 6)
            out[f'nmask{b}'] = np.int64(sum(len(m) for m in segm_result))
    out['panoptic_seg'] = np.stack(segs).astype(np.int32)
    out['nseg'] = np.array(nseg, dtype=np.int64)
    np.savez_compressed(os.path.join(OUT, name + '.npz'), **out)
    print(f'{name}: ok  segments per frame = {nseg}  void = {float((np.stack(segs) == 0).mean()):.3f}')


def run_pan_case(name, p):
    """KernelIterHead.get_panoptic of the reference (merge_joint=True) on structured synthetic logits."""
    test_cfg = AttrDict(max_per_img=p['Np'], mask_thr=0.5, stuff_score_thr=0.05,
                        merge_stuff_thing=AttrDict(overlap_thr=0.6, iou_thr=0.5, stuff_max_area=4096, instance_score_thr=0.25))
    cfg = head_cfg(False, C=32, heads=8, ffn=64, ncls=p['ncls'], n_thing=p['T'], n_stuff=p['ncls'] - p['T'], S=1, up=p['up'],
                   nprop=p['Np'])
    cfg.update(do_panoptic=True, merge_joint=True, test_cfg=test_cfg)
    head = build_head(cfg)
    head.eval()
    cls, logits = (torch.from_numpy(a) for a in synth.panoptic_inputs(p['B'], p['N'], p['Np'], p['ncls'], p['Hm'], p['Wm'], p['seed']))
    meta = dict(img_shape=(*p['img'], 3), batch_input_shape=tuple(p['bis']), ori_shape=(*p['ori'], 3))
    segs, infos = [], []
    with torch.no_grad():
        scaled = logits
        if p['up'] > 1:   # the last stage's upsample, exactly as _mask_forward does it (knet/det/kernel_iter_head.py:122-130)
            scaled = F.interpolate(logits, scale_factor=p['up'], align_corners=False, mode='bilinear')
        for b in range(p['B']):
            _, _, (pan, info) = head.get_panoptic(cls[b], scaled[b], head.test_cfg, meta)
            segs.append(pan)
            # id, isthing, category_id, instance_id (-1 for stuff), score (nan for stuff), area (-1 for things)
            infos.append(np.array([[s['id'], int(s['isthing']), s['category_id'], s.get('instance_id', -1),
                                    s.get('score', float('nan')), s.get('area', -1)] for s in info], dtype=np.float64).reshape(-1, 6))
    out = dict(case=np.array([p['B'], p['N'], p['Np'], p['T'], p['ncls'], p['Hm'], p['Wm'], p['up'], *p['bis'], *p['img'], *p['ori'],
                              p['seed']], dtype=np.int64),
               panoptic_seg=np.stack(segs).astype(np.int32), nseg=np.array([len(i) for i in infos], dtype=np.int64))
    for b, i in enumerate(infos):
        out[f'info{b}'] = i
    np.savez_compressed(os.path.join(OUT, name + '.npz'), **out)
    print(f'{name}: ok  segments per frame = {[len(i) for i in infos]}  void fraction = '
          f'{float((np.stack(segs) == 0).mean()):.3f}')


ASSIGN_CASES = {
    'assign_tiny': dict(N=12, G=5, ncls=2, H=16, W=32, seed=51),
    'assign_cfg': dict(N=100, G=23, ncls=2, H=64, W=128, seed=52),       # 100 thing kernels, the shipped costs, 1/4-res masks
    'assign_odd': dict(N=21, G=30, ncls=7, H=9, W=15, seed=53),          # more ground truths than kernels, ragged P
}


def run_assign_case(name, p):
    """MaskHungarianAssigner.assign of the reference with the shipped train_cfg costs
    (configs/det/_base_/models/knet_kitti_step_s3_r50_fpn.py:143-148)."""
    from mmdet.core import build_assigner
    assigner = build_assigner(dict(type='MaskHungarianAssigner', cls_cost=dict(type='FocalLossCost', weight=2.0),
                                   dice_cost=dict(type='DiceCost', weight=4.0, pred_act=True),
                                   mask_cost=dict(type='MaskCost', weight=1.0, pred_act=True)))
    logits, cls, gt, labels = (torch.from_numpy(a) for a in synth.assign_inputs(p['N'], p['G'], p['ncls'], p['H'], p['W'], p['seed']))
    with torch.no_grad():
        res = assigner.assign(logits, cls, gt, labels)
        # the cost matrix itself, through the reference's own cost objects (what assign() sums at :222-241)
        cost = assigner.cls_cost(cls, labels) + assigner.mask_cost(logits, gt) + assigner.dice_cost(logits, gt)
    np.savez_compressed(os.path.join(OUT, name + '.npz'),
                        case=np.array([p['N'], p['G'], p['ncls'], p['H'], p['W'], p['seed']], dtype=np.int64),
                        gt_inds=res.gt_inds.numpy(), labels=res.labels.numpy(), cost=cost.numpy())
    print(f'{name}: ok  matched {int((res.gt_inds > 0).sum())} of {p["N"]} kernels to {p["G"]} ground truths')


TRAIN_CASES = {
    # det head, panoptic targets (thing gt + stuff sem targets), soft-edged gt masks
    'train_tiny': dict(video=False, C=64, heads=8, ffn=128, ncls=5, n_thing=2, n_stuff=3, S=3, up=2, nprop=12, N=15, H=8, W=16, B=2,
                       seed=71),
    # video head: last-stage link to the previous frame's kernels (forward_train_with_previous), x4 upsample
    'train_video': dict(video=True, C=64, heads=8, ffn=128, ncls=5, n_thing=2, n_stuff=3, S=2, up=4, nprop=12, N=15, H=8, W=16, B=2,
                        seed=72),
    # the "update" video head in training: previous_link='update_dynamic_cov' rewrites the last stage's kernels, previous_type='update'
    'train_video_upd': dict(video=True, C=64, heads=8, ffn=128, ncls=5, n_thing=2, n_stuff=3, S=2, up=4, nprop=12, N=15, H=8, W=16, B=2,
                            seed=74, plink='update_dynamic_cov', ptype='update'),
    # config channels / kernel count
    'train_cfg': dict(video=False, C=256, heads=8, ffn=2048, ncls=19, n_thing=2, n_stuff=17, S=3, up=2, nprop=100, N=117, H=16, W=32,
                      B=2, seed=73),
    # video head at config channels / kernel count, x4 (the shipped KITTI-STEP video config: mask_upsample_stride=4,
    # configs/det/video_knet_kitti_step/...link_ffn_joint_train.py:102), full gradients w.r.t. x and the kernels
    'train_video_c256': dict(video=True, C=256, heads=8, ffn=2048, ncls=19, n_thing=2, n_stuff=17, S=3, up=4, nprop=100, N=117, H=16,
                             W=32, B=2, seed=75, full_x=True),
    # BASELINE cfg3 at the size bench.py --train times: 1024x2048 frames -> 128x256 features, 512x1024 loss masks, two frames
    'train_video_cfg3': dict(video=True, C=256, heads=8, ffn=2048, ncls=19, n_thing=2, n_stuff=17, S=3, up=4, nprop=100, N=117, H=128,
                             W=256, B=2, seed=76),
}
TRAIN_GRAD_KEYS = ('mask_head.0.feat_transform.conv.weight', 'mask_head.0.feat_transform.conv.bias',
                   'mask_head.0.kernel_update_conv.dynamic_layer.weight', 'mask_head.0.attention.attn.in_proj_weight',
                   'mask_head.0.ffn.layers.1.weight', 'mask_head.1.fc_mask.weight', 'mask_head.1.fc_cls.bias',
                   'mask_head.1.kernel_update_conv.fc_norm.weight', 'mask_head.1.mask_fcs.0.weight')


def _train_step(p, N, H, W, B, seed, dtype=torch.float32):
    """One training step of the reference head of case `p` in `dtype` -> (head, losses, track, total, assigned, x, pf).  float64 is the
    reference's OWN code evaluated in double precision (weights / inputs / targets widened; its hard-coded `.float()` of the binarised
    mask — knet/det/kernel_update_head.py:192 — widened at the einsum that consumes it): the tie-breaker for gradient rows where the
    fp32 evaluation sits on a ReLU kink (`grad_*_f64` below)."""
    video = p['video']
    cfg = head_cfg(**p)
    cfg['train_cfg'] = [AttrDict(assigner=dict(type='MaskHungarianAssigner', cls_cost=dict(type='FocalLossCost', weight=2.0),
                                               dice_cost=dict(type='DiceCost', weight=4.0, pred_act=True),
                                               mask_cost=dict(type='MaskCost', weight=1.0, pred_act=True)),
                                 sampler=dict(type='MaskPseudoSampler'), pos_weight=1) for _ in range(p['S'])]
    head = build_head(cfg)
    head.train()
    shapes = {k: tuple(v.shape) for k, v in head.state_dict().items()}
    load_formula_weights(head, shapes, seed)
    head = head.to(dtype)
    x, pf, mp = (torch.from_numpy(a).to(dtype) for a in synth.head_inputs(B, N, p['C'], H, W, seed))
    x.requires_grad_(True)
    pf.requires_grad_(True)
    tg = synth.train_targets(B, p['n_thing'], p['n_stuff'], H * p['up'], W * p['up'], seed)
    gt_masks = [torch.from_numpy(t['gt_masks']).to(dtype) for t in tg]
    gt_labels = [torch.from_numpy(t['gt_labels']) for t in tg]
    gt_sem_seg = [torch.from_numpy(t['gt_sem_seg']).to(dtype) for t in tg]
    gt_sem_cls = [torch.from_numpy(t['gt_sem_cls']) for t in tg]
    metas = [dict() for _ in range(B)]
    # record the assignment of every stage
    assigned = []
    for a in head.mask_assigner:
        orig = a.assign

        def rec(*args, _orig=orig, **kw):
            r = _orig(*args, **kw)
            assigned.append(r.gt_inds.clone())
            return r
        a.assign = rec
    einsum = torch.einsum
    if dtype == torch.float64:
        torch.einsum = lambda eq, *ops: einsum(eq, *[o.double() for o in ops])
    try:
        if video:
            prev = torch.from_numpy(synth.normalish((B, N, p['C'], 1, 1), 99 + seed, 1.0)).to(dtype)
            out = head.forward_train_with_previous(x, pf, mp, None, metas, gt_masks, gt_labels, gt_sem_seg=gt_sem_seg,
                                                   gt_sem_cls=gt_sem_cls, previous_obj_feats=prev)
            losses, track = out[0], out[5]
        else:
            losses = head.forward_train(x, pf, mp, None, metas, gt_masks, gt_labels, gt_sem_seg=gt_sem_seg, gt_sem_cls=gt_sem_cls)
            track = None
        total = sum(v for k, v in losses.items() if 'loss' in k)
        if track is not None:
            total = total + 0.01 * (track ** 2).sum()     # makes the link's parameters part of the graph
        total.backward()
    finally:
        torch.einsum = einsum
    return head, losses, track, total, assigned, x, pf


def run_train_case(name, p):
    """`forward_train` / `forward_train_with_previous` of the reference (knet/det/kernel_iter_head.py:139-231,
    knet/video/kernel_iter_head.py:255-376) with the shipped train_cfg: per-stage losses, the assignments, and the gradients
    of the summed loss w.r.t. x, proposal_feats and a sample of the parameters."""
    p = dict(p)
    N, H, W, B, seed = (p.pop(k) for k in ('N', 'H', 'W', 'B', 'seed'))
    full_x = p.pop('full_x', False)
    video = p['video']
    head, losses, track, total, assigned, x, pf = _train_step(p, N, H, W, B, seed)
    out = dict(case=np.array([p['C'], p['heads'], p['ffn'], p['ncls'], p['n_thing'], p['n_stuff'], p['S'], p['up'], p['nprop'], N, H,
                              W, B, seed, int(video)], dtype=np.int64),
               loss_keys=np.array(sorted(losses)), loss_vals=np.array([float(losses[k]) for k in sorted(losses)], dtype=np.float64),
               total=np.float64(float(total)), assigned=torch.stack(assigned).numpy())
    if p.get('plink') is not None or p.get('ptype', 'ffn') != 'ffn':
        out['plink'], out['ptype'] = np.array(p['plink'] or ''), np.array(p['ptype'])
    big = p['C'] > 64

    def put(tag, t, full=False):
        """full tensor for the small cases; 4096 sampled elements + the norm for the config-size case"""
        t = t.detach()
        if full or not big or t.numel() <= 8192:
            out[tag] = t.numpy()
        else:
            idx = (synth.uniform((4096,), 5151 + len(tag), 0.0, 1.0).astype(np.float64) * t.numel()).astype(np.int64)
            out[tag + '_idx'], out[tag + '_val'] = idx, t.reshape(-1)[idx].numpy()
            out[tag + '_norm'] = np.float64(float(t.double().norm()))
    put('grad_x', x.grad, full_x)
    put('grad_pf', pf.grad, full_x)
    if full_x:
        # the same step evaluated in float64 (see _train_step): where the fp32 evaluation's gradient of a kernel row differs from this
        # by percents while every other row agrees to 1e-4, the fp32 run sat on a ReLU kink (a pre-activation within rounding of 0)
        _, _, _, _, assigned64, x64, pf64 = _train_step(p, N, H, W, B, seed, torch.float64)
        assert all(torch.equal(a, b) for a, b in zip(assigned, assigned64))
        out['grad_x_f64'], out['grad_pf_f64'] = x64.grad.float().numpy(), pf64.grad.float().numpy()
    named = dict(head.named_parameters())
    gk = [k for k in TRAIN_GRAD_KEYS if k in named]
    if video and p.get('plink') is not None:
        gk += ['mask_head.%d.%s' % (p['S'] - 1, k) for k in (
            'attention_previous_update_link.dynamic_layer.weight', 'attention_previous_link.attn.in_proj_weight',
            'link_ffn_link.layers.1.weight', 'attention_previous_update_track.fc_layer.weight',
            'attention_previous_track.attn.out_proj.weight', 'link_ffn_norm_track.weight')]
    elif video:
        gk += ['mask_head.%d.attention_previous.attn.in_proj_weight' % (p['S'] - 1), 'mask_head.%d.link_ffn.layers.1.weight' % (p['S'] - 1)]
    out['grad_keys'] = np.array(gk)
    for i, k in enumerate(gk):
        put(f'grad_{i}', named[k].grad)
    # every parameter's gradient norm (cheap completeness check)
    out['all_keys'] = np.array(sorted(named))
    out['all_gnorm'] = np.array([float(named[k].grad.double().norm()) if named[k].grad is not None else -1.0 for k in sorted(named)])
    if track is not None:
        out['track'] = track.detach().numpy()
    np.savez_compressed(os.path.join(OUT, name + '.npz'), **out)
    print(f'{name}: ok  total={float(total):.5f}  ' + ' '.join(f'{k}={float(v):.4f}' for k, v in sorted(losses.items())[:6]))


RPN_TRAIN_CASES = {
    # the shipped rpn_head losses / train_cfg.rpn (configs/det/_base_/models/knet_kitti_step_s3_r50_fpn.py:29-78, 143-150) behind the
    # pass-through neck: x2 up-scaled predictions for assignment and losses, stuff kernels appended in training
    'rpn_train_tiny': dict(C=64, nprop=12, ncls=5, n_thing=2, H=8, W=16, B=2, seed=91),
    'rpn_train_cfg': dict(C=256, nprop=100, ncls=19, n_thing=2, H=16, W=32, B=2, seed=92),
}


def run_rpn_train_case(name, p):
    """`ConvKernelHead.forward_train` of the reference (knet/det/kernel_head.py:267-336): losses, assignments, outputs handed to the
    roi head, gradients of the summed loss w.r.t. both feature maps and every parameter."""
    cfg = dict(type='ConvKernelHead', num_proposals=p['nprop'], in_channels=p['C'], out_channels=p['C'], num_loc_convs=0,
               num_seg_convs=0, localization_fpn=dict(type='PassThroughNeck'), conv_kernel_size=1, semantic_fpn=True,
               num_classes=p['ncls'], use_binary=True, proposal_feats_with_obj=True, feat_downsample_stride=2, feat_refine=False,
               num_thing_classes=p['n_thing'], num_stuff_classes=p['ncls'] - p['n_thing'], cat_stuff_mask=True,
               loss_rank=dict(type='CrossEntropyLoss', use_sigmoid=False, loss_weight=0.1),
               loss_seg=dict(type='FocalLoss', use_sigmoid=True, gamma=2.0, alpha=0.25, loss_weight=1.0),
               loss_mask=dict(type='CrossEntropyLoss', use_sigmoid=True, loss_weight=1.0),
               loss_dice=dict(type='DiceLoss', loss_weight=4.0),
               train_cfg=AttrDict(assigner=dict(type='MaskHungarianAssigner', cls_cost=dict(type='FocalLossCost', weight=2.0),
                                                dice_cost=dict(type='DiceCost', weight=4.0, pred_act=True),
                                                mask_cost=dict(type='MaskCost', weight=1.0, pred_act=True)),
                                  sampler=dict(type='MaskPseudoSampler'), pos_weight=1))
    head = build_head(cfg)
    head.train()
    loc, sem, shapes = init_inputs(dict(p, sem=True))
    assert {k: tuple(v.shape) for k, v in head.state_dict().items()} == shapes
    head.load_state_dict({k: torch.from_numpy(v) for k, v in synth.state_dict_like(shapes, p['seed']).items()}, strict=True)
    loc, sem = torch.from_numpy(loc).requires_grad_(True), torch.from_numpy(sem).requires_grad_(True)
    tg = synth.train_targets(p['B'], p['n_thing'], p['ncls'] - p['n_thing'], 2 * p['H'], 2 * p['W'], p['seed'])
    gt_masks = [torch.from_numpy(t['gt_masks']) for t in tg]
    gt_labels = [torch.from_numpy(t['gt_labels']) for t in tg]
    gt_sem_seg = [torch.from_numpy(t['gt_sem_seg']) for t in tg]
    gt_sem_cls = [torch.from_numpy(t['gt_sem_cls']) for t in tg]
    assigned = []
    orig = head.assigner.assign

    def rec(*args, **kw):
        r = orig(*args, **kw)
        assigned.append(r.gt_inds.clone())
        return r
    head.assigner.assign = rec
    losses, prop, x_feats, masks, cls = head.forward_train((loc, sem), [dict() for _ in range(p['B'])], gt_masks, gt_labels,
                                                           gt_sem_seg=gt_sem_seg, gt_sem_cls=gt_sem_cls)
    assert cls is None
    total = sum(v for k, v in losses.items() if 'loss' in k) + 1e-3 * (prop ** 2).mean() + 1e-3 * (masks ** 2).mean()
    total.backward()
    named = dict(head.named_parameters())
    out = dict(case=np.array([p['C'], p['nprop'], p['ncls'], p['n_thing'], p['H'], p['W'], p['B'], p['seed'], 1, 1], dtype=np.int64),
               loss_keys=np.array(sorted(losses)), loss_vals=np.array([float(losses[k].detach()) for k in sorted(losses)], dtype=np.float64),
               total=np.float64(float(total.detach())), assigned=torch.stack(assigned).numpy(),
               proposal_feats=prop.detach().numpy(), mask_rowsum=masks.detach().double().sum(dim=(-1, -2)).numpy(),
               grad_keys=np.array(sorted(named)))
    big = p['C'] > 64
    for tag, t in [('grad_loc', loc.grad), ('grad_sem', sem.grad)] + [(f'grad_{i}', named[k].grad) for i, k in enumerate(sorted(named))]:
        if not big or t.numel() <= 8192:
            out[tag] = t.numpy()
        else:
            idx = (synth.uniform((4096,), 6161 + len(tag), 0.0, 1.0).astype(np.float64) * t.numel()).astype(np.int64)
            out[tag + '_idx'], out[tag + '_val'] = idx, t.reshape(-1)[idx].numpy()
            out[tag + '_norm'] = np.float64(float(t.double().norm()))
    np.savez_compressed(os.path.join(OUT, name + '.npz'), **out)
    print(f'{name}: ok  total={float(total):.5f}  ' + ' '.join(f'{k}={float(v):.4f}' for k, v in sorted(losses.items())))


def run_instance_case():
    """Instance-only results (do_panoptic=False): KernelIterHead.simple_test -> top-k -> get_seg_masks / segm2result
    (knet/det/kernel_iter_head.py:270-281, knet/det/kernel_update_head.py:443-481), YouTube-VIS-like class layout (things only)."""
    p = dict(video=False, C=64, heads=8, ffn=128, ncls=7, n_thing=7, n_stuff=0, S=2, up=2, nprop=20)
    N, H, W, B, seed = 20, 8, 16, 2, 81
    cfg = head_cfg(**p)
    cfg.update(do_panoptic=False, test_cfg=AttrDict(max_per_img=10, mask_thr=0.5))
    head = build_head(cfg)
    head.eval()
    shapes = {k: tuple(v.shape) for k, v in head.state_dict().items()}
    load_formula_weights(head, shapes, seed)
    x, pf, mp = (torch.from_numpy(a) for a in synth.head_inputs(B, N, p['C'], H, W, seed))
    meta = dict(img_shape=(60, 120, 3), batch_input_shape=(64, 128), ori_shape=(90, 180, 3))
    with torch.no_grad():
        res = head.simple_test(x, pf, mp, None, [meta] * B)
    out = dict(case=np.array([p['C'], p['heads'], p['ffn'], p['ncls'], p['n_thing'], p['n_stuff'], p['S'], p['up'], p['nprop'], N, H, W,
                              B, seed, 0], dtype=np.int64))
    for b, (bbox_result, segm_result) in enumerate(res):
        out[f'scores{b}'] = np.concatenate([bb[:, 4] for bb in bbox_result])            # class-major, score order within a class
        out[f'labels{b}'] = np.concatenate([np.full(len(bb), c) for c, bb in enumerate(bbox_result)]).astype(np.int64)
        masks = [m for per_cls in segm_result for m in per_cls]
        out[f'masks{b}'] = np.packbits(np.stack(masks).astype(bool)) if masks else np.zeros(0, np.uint8)
        out[f'nmask{b}'] = np.int64(len(masks))
    np.savez_compressed(os.path.join(OUT, 'inst_tiny.npz'), **out)
    print('inst_tiny: ok  instances per image =', [int(out[f'nmask{b}']) for b in range(B)])


def run_assign_soft():
    """MaskHungarianAssigner with SOFT ground-truth masks (bilinearly down-sampled, knet/det/knet.py:131): the costs use the real
    values of the targets, not their binarisation."""
    from mmdet.core import build_assigner
    assigner = build_assigner(dict(type='MaskHungarianAssigner', cls_cost=dict(type='FocalLossCost', weight=2.0),
                                   dice_cost=dict(type='DiceCost', weight=4.0, pred_act=True),
                                   mask_cost=dict(type='MaskCost', weight=1.0, pred_act=True)))
    N, G, ncls, H, W, seed = 30, 9, 3, 32, 64, 54
    logits, cls, gt, labels = (torch.from_numpy(a) for a in synth.assign_inputs(N, G, ncls, 2 * H, 2 * W, seed))
    logits = F.interpolate(logits[None], size=(H, W), mode='bilinear', align_corners=False)[0]
    gt = F.interpolate(gt[None], size=(H, W), mode='bilinear', align_corners=False)[0]      # soft borders: values in {0, .25, .5, .75, 1}
    with torch.no_grad():
        res = assigner.assign(logits, cls, gt, labels)
        cost = assigner.cls_cost(cls, labels) + assigner.mask_cost(logits, gt) + assigner.dice_cost(logits, gt)
    np.savez_compressed(os.path.join(OUT, 'assign_soft.npz'), case=np.array([N, G, ncls, H, W, seed], dtype=np.int64),
                        gt_inds=res.gt_inds.numpy(), labels=res.labels.numpy(), cost=cost.numpy(),
                        soft_fraction=np.float64(float(((gt > 0) & (gt < 1)).float().mean())))
    print(f'assign_soft: ok  soft pixels {float(((gt > 0) & (gt < 1)).float().mean()):.3f}')


def thr_kat():
    """(sigmoid(z) > 0.5) as the reference computes it (knet/det/kernel_update_head.py:190-191) — torch CPU fp32.
    The flip point is not z=0: it depends on the fp32 sigmoid (SURVEY.md §7 'Threshold semantics')."""
    base = np.array([0.0, 5e-8, -5e-8, 8.9e-8, 8.94e-8, 9e-8, 1e-7, -1e-7, 1.2e-7, 2e-7, -2e-7, 1.0, -1.0, 1e-3, -1e-3,
                     float('inf'), float('-inf'), 30.0, -30.0, 100.0, -100.0], dtype=np.float32)
    # dense sweep of every fp32 value in a window around the flip point
    lo = np.float32(0.0).view(np.uint32)
    sweep = (np.arange(0, 1 << 12, dtype=np.uint32) * np.uint32(1 << 9) + np.float32(1e-8).view(np.uint32)).view(np.float32)
    near = np.float32(8.94e-8).view(np.uint32).astype(np.int64) + np.arange(-4096, 4096)
    near = near.astype(np.uint32).view(np.float32)
    z = np.concatenate([base, sweep, near, -sweep[:512]])
    del lo
    # exercise both the vectorised body and scalar tail of ATen's CPU sigmoid: pad to odd length, two layouts
    t = torch.from_numpy(z)
    a = (t.sigmoid() > 0.5).numpy()
    b = (t.reshape(-1, 1).expand(-1, 3).contiguous().sigmoid() > 0.5)[:, 1].numpy()
    assert (a == b).all()
    pos = z[a]
    flip = float(pos[pos > 0].min()) if (pos > 0).any() else float('nan')
    np.savez_compressed(os.path.join(OUT, 'thr_kat.npz'), z=z, bit=a, flip=np.float32(flip))
    print(f'thr_kat: smallest fp32 z with sigmoid(z)>0.5 in this sweep = {flip!r}')


if __name__ == '__main__':
    torch.manual_seed(0)
    torch.set_num_threads(8)
    os.makedirs(OUT, exist_ok=True)
    only = sys.argv[1:]
    for name, p in CASES.items():
        if not only or name in only:
            run_case(name, p)
    for name, p in INIT_CASES.items():
        if not only or name in only:
            run_init_case(name, p)
    for name, p in PAN_CASES.items():
        if not only or name in only:
            run_pan_case(name, p)
    for nm, src in (('pan_tf_tiny', 'pan_tiny'), ('pan_tf_cfg', 'pan_cfg')):
        if not only or nm in only:
            run_pan_thing_first(nm, PAN_CASES[src])
    if not only or 'pan_video' in only:
        run_pan_video_case('pan_video', PAN_CASES['pan_tiny'])
    for name, p in ASSIGN_CASES.items():
        if not only or name in only:
            run_assign_case(name, p)
    for name, p in TRAIN_CASES.items():
        if not only or name in only:
            run_train_case(name, p)
    for name, p in RPN_TRAIN_CASES.items():
        if not only or name in only:
            run_rpn_train_case(name, p)
    if not only or 'assign_soft' in only:
        run_assign_soft()
    if not only or 'inst_tiny' in only:
        run_instance_case()
    if not only or 'init_keys' in only:
        init_keys()
    if not only or 'thr_kat' in only:
        thr_kat()

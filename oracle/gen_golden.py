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

from oracle import synth  # noqa: E402

import knet.kernel_updator  # noqa: E402,F401  (registers KernelUpdator)
import knet.det.kernel_update_head  # noqa: E402,F401
import knet.det.kernel_iter_head  # noqa: E402,F401
import knet.video.kernel_update_head  # noqa: E402,F401
import knet.video.kernel_iter_head  # noqa: E402,F401
from mmdet.models.builder import build_head  # noqa: E402

OUT = os.path.join(ROOT, 'tests', 'golden')


def head_cfg(video, C, heads, ffn, ncls, n_thing, n_stuff, S, up, nprop):
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
        mh.update(previous='placeholder', previous_type='ffn')
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
        else:
            o, c, m, sc = head.simple_test_mask_preds(x, pf, mp, None, metas)
    assert torch.equal(per_stage[-1][1], m)
    big = name.endswith('_big')
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
    if not only or 'thr_kat' in only:
        thr_kat()

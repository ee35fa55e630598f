"""CPU (-m "not gpu"): registry / config build, state-dict drop-in, C-ABI export surface, host-side helpers.
No compute kernel is launched here."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

from helpers import GOLDEN, load_golden

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _cfg(video, **kw):
    """The roi_head dict of the shipped det / video configs (video-k-net_amd/configs.py), train_cfg=None."""
    import vkn_import
    return vkn_import.load().configs.roi_head_cfg(video, **kw)


@pytest.mark.parametrize('video', [False, True])
def test_reference_config_builds_and_state_dict_matches_reference(vkn, video):
    head = vkn.build_head(_cfg(video))
    assert type(head).__name__ == ('VideoKernelIterHead' if video else 'KernelIterHead')
    g, _ = load_golden('video_cfg' if video else 'det_cfg')
    sd = head.state_dict()
    assert sorted(sd) == list(g['keys'])                                   # same 44(+16) keys per stage as the reference
    assert [str(tuple(sd[k].shape)) for k in sorted(sd)] == list(g['shapes'])
    head.init_weights()
    assert float(head.mask_head[0].fc_cls.bias.detach()[0]) == pytest.approx(-np.log(99.0), rel=1e-6)   # bias_init_with_prob(0.01)
    # a reference-shaped checkpoint loads strictly
    head.load_state_dict({k: torch.zeros(tuple(v.shape)) for k, v in sd.items()}, strict=True)


@pytest.mark.parametrize('name', ['video_upd_tiny', 'video_updffn_tiny', 'video_upd_cfg', 'video_latt_upd_tiny', 'video_updobj_tiny',
                                  'video_latt_updobj_tiny'])
def test_update_link_configs_build_and_state_dict_matches_reference(vkn, name):
    """previous_link='update_dynamic_cov' + previous_type='update' | 'ffn' (three shipped swin configs) and the combinations nobody
    ships (`link_atten`, `update_obj`): same module tree / keys."""
    g, case = load_golden(name)
    head = vkn.build_head(_cfg(True, C=case['C'], heads=case['heads'], ffn=case['ffn'], ncls=case['ncls'], n_thing=case['n_thing'],
                               n_stuff=case['n_stuff'], S=case['S'], up=case['up'], nprop=case['nprop'],
                               mask_over=dict(previous_link=case['plink'], previous_type=case['ptype'])))
    sd = head.state_dict()
    assert sorted(sd) == list(g['keys'])
    assert [str(tuple(sd[k].shape)) for k in sorted(sd)] == list(g['shapes'])
    last = head.mask_head[-1]
    if case['plink'] is not None:   # the pre-link's updator exists only for `update_dynamic_cov` (`link_atten`: attention + FFN only)
        assert last._link_names('link')[0] == ('attention_previous_update_link' if case['plink'] == 'update_dynamic_cov' else None)
    assert (last._link_names('track')[0] is None) == (case['ptype'] == 'ffn')


def test_registry_surface(vkn):
    for name in ('KernelIterHead', 'KernelUpdateHead', 'VideoKernelIterHead', 'VideoKernelUpdateHead'):
        assert vkn.HEADS.get(name) is not None
    assert vkn.TRANSFORMER_LAYER.get('KernelUpdator') is vkn.KernelUpdator
    ku = vkn.build_transformer_layer(dict(type='KernelUpdator', in_channels=64, feat_channels=64, out_channels=64))
    assert sorted(n for n, _ in ku.named_parameters())[0] == 'dynamic_layer.bias'
    with pytest.raises(KeyError):
        vkn.build_head(dict(type='NoSuchHead'))
    # feat_transform_cfg.pop('kernel_size') mutates the caller's dict exactly like the reference (kernel_update_head.py:108)
    cfg = _cfg(False, C=64, ffn=128, S=1)
    cfg['mask_head'][0]['feat_transform_cfg']['kernel_size'] = 1
    vkn.build_head(cfg)
    assert 'kernel_size' not in cfg['mask_head'][0]['feat_transform_cfg']


def test_unsupported_variants_fail_loudly(vkn):
    bad = _cfg(False, C=64, ffn=128, S=1)
    bad['mask_head'][0]['conv_kernel_size'] = 3
    with pytest.raises(NotImplementedError):
        vkn.build_head(bad)
    bad = _cfg(True, C=64, ffn=128, S=1)
    bad['mask_head'][0]['previous_type'] = 'no_such_link'
    with pytest.raises(ValueError):
        vkn.build_head(bad)
    with pytest.raises(NotImplementedError):
        vkn.build_head(dict(_cfg(False, C=64, ffn=128, S=1), train_cfg=[dict(assigner=dict(type='MaskHungarianAssigner'))]))


def test_no_cpu_fallback(vkn):
    head = vkn.build_head(_cfg(False, C=64, heads=8, ffn=128, ncls=5, S=1)).eval()
    x, pf, mp = torch.zeros(1, 64, 4, 8), torch.zeros(1, 6, 64, 1, 1), torch.zeros(1, 6, 4, 8)
    with pytest.raises(vkn.VknLibraryError):
        head.simple_test_mask_preds(x, pf, mp, None, [dict()])
    with pytest.raises(vkn.VknLibraryError):
        vkn.ops.mask_gather(x, mp)
    with pytest.raises(vkn.VknLibraryError):
        vkn.ops.mask_decode(x, pf)


def test_library_exports_every_declared_symbol(vkn):
    hdr = open(os.path.join(ROOT, 'include', 'vkn.h')).read()
    declared = sorted(set(re.findall(r'\b(vkn_[a-z0-9_]+)\s*\(', hdr)))
    assert declared == sorted(vkn._lib.SYMBOLS)
    L = ctypes.CDLL(vkn._lib.LIBPATH)
    for sym in declared:
        assert getattr(L, sym) is not None
    L2 = vkn._lib.lib()
    assert L2.vkn_version() == 0x000600
    assert L2.vkn_strerror(0) == b'ok' and b'workspace' in L2.vkn_strerror(-3)
    assert L2.vkn_sizeof_dims() == ctypes.sizeof(vkn._lib.VknDims)
    assert L2.vkn_sizeof_stage_weights() == ctypes.sizeof(vkn._lib.VknStageWeights)
    assert L2.vkn_sizeof_split_item() == ctypes.sizeof(vkn._lib.VknSplitItem)      # the batched training-chain tables (round 4)
    assert L2.vkn_sizeof_dw_item() == ctypes.sizeof(vkn._lib.VknDwItem)


def test_workspace_queries_are_pure_host(vkn):
    L = vkn._lib.lib()
    d = vkn.ops.make_dims(8, 117, 256, 128, 256, 8, 2048, 19, 1, 1)
    stage, head = L.vkn_stage_workspace_bytes(ctypes.byref(d)), L.vkn_head_workspace_bytes(ctypes.byref(d))
    assert 0 < stage < head < (1 << 31)
    assert head - stage >= 8 * 117 * 128 * 256 * 4                         # the mask ping-pong buffer
    bad = vkn.ops.make_dims(1, 117, 250, 8, 8, 8, 2048, 19, 1, 1)          # C % 32 != 0
    assert L.vkn_stage_workspace_bytes(ctypes.byref(bad)) == 0
    assert L.vkn_gather_workspace_bytes(8, 117, 256, 32768) > 0 and L.vkn_decode_workspace_bytes(8, 117, 256) > 0
    # argument errors are reported, never crash (null pointers, no GPU touched)
    assert L.vkn_mask_gather_f32(None, None, 0.0, None, None, 1, 1, 32, 32, None, 0, 0, None) == -1
    assert L.vkn_mask_decode_f32(None, None, None, None, 1, 1, 32, 32, None, 0, 0, None) == -1
    assert L.vkn_upsample_bilinear_f32(None, None, 1, 1, 1, 2, None) == -1
    assert L.vkn_stage_forward_f32(ctypes.byref(bad), None, *([None] * 9), None, 0, 0, None) == -2


def test_threshold_logit_matches_reference_flip_point(vkn):
    g = np.load(os.path.join(GOLDEN, 'thr_kat.npz'))
    t = vkn.ops.thr_logit(0.5)
    assert np.float32(t) == np.float32(g['flip'])                          # 8.940697e-08, not 0
    assert np.array_equal(g['z'] >= np.float32(t), g['bit'])               # z >= thr_logit  <=>  sigmoid(z) > 0.5 (bit-exact)
    # other thresholds: bisection agrees with torch on a dense neighbourhood
    for thr in (0.3, 0.7):
        tz = np.float32(vkn.ops.thr_logit(thr))
        z = (tz.view(np.uint32).astype(np.int64) + np.arange(-64, 64)).astype(np.uint32).view(np.float32)
        assert np.array_equal(z >= tz, (torch.from_numpy(z).sigmoid() > thr).numpy())


@pytest.mark.parametrize('tag,refine', [('kitti', False), ('refine', True)])
def test_conv_kernel_head_state_dict_matches_reference(vkn, tag, refine):
    """ConvKernelHead under the shipped rpn_head kwargs (configs/det/_base_/models/knet_kitti_step_s3_r50_fpn.py:29-61): same
    state-dict keys and shapes as the reference (captured by oracle/gen_golden.py:init_keys)."""
    g = dict(np.load(os.path.join(GOLDEN, 'init_keys.npz'), allow_pickle=False))
    head = vkn.build_head(dict(type='ConvKernelHead', num_classes=19, num_thing_classes=2, num_stuff_classes=17,
                               cat_stuff_mask=True, conv_kernel_size=1, feat_downsample_stride=2, feat_refine_stride=1,
                               feat_refine=refine, use_binary=True, num_loc_convs=1, num_seg_convs=1, conv_normal_init=True,
                               localization_fpn=None, num_proposals=100, proposal_feats_with_obj=True,
                               xavier_init_kernel=False, kernel_init_std=1))
    sd = head.state_dict()
    assert sorted(sd) == list(g[tag + '_keys'])
    assert [str(tuple(sd[k].shape)) for k in sorted(sd)] == list(g[tag + '_shapes'])
    head.init_weights()
    assert vkn.build_head(dict(type='ConvKernelHead', proposal_feats_with_obj=True, use_binary=False)).use_binary is False   # soft weights: built
    with pytest.raises(NotImplementedError):
        vkn.build_head(dict(type='ConvKernelHead', conv_kernel_size=3))
    with pytest.raises(vkn.VknLibraryError):   # CPU tensors: no fallback
        head.eval().decode_init_proposals_from_feats(torch.zeros(1, 256, 4, 8), torch.zeros(1, 256, 4, 8))


def test_widened_entry_points_validate_on_the_host(vkn):
    """Argument / shape validation of the kernel-init and panoptic entry points happens before any launch (no GPU needed)."""
    L = vkn._lib.lib()
    assert L.vkn_kernel_init_workspace_bytes(2, 100, 19, 256, 32768) > 0
    assert L.vkn_kernel_init_workspace_bytes(0, 100, 19, 256, 32768) == 0
    cfg = vkn._lib.VknPanopticCfg(100, 2, 100, 0.25, 0.6, 4, 128, 256, 1024, 2048, 1024, 2048, 1024, 2048)
    assert L.vkn_sizeof_panoptic_cfg() == ctypes.sizeof(cfg)
    nb = L.vkn_panoptic_workspace_bytes(ctypes.byref(cfg), 8, 117)
    assert nb >= 8 * 117 * 7 * 4
    # null pointers / bad geometry are rejected with an error code, never a crash
    assert L.vkn_panoptic_joint_f32(ctypes.byref(cfg), None, None, 8, 117, 19, None, None, None, None, None, 0, None) == -1
    assert L.vkn_kernel_init_f32(None, None, None, None, None, 2, 1, 1, 0.0, None, None, None, None, 2, 100, 19, 256, 1024,
                                 None, 0, 0, None) == -1
    assert vkn._lib.lib().vkn_strerror(-2).decode().startswith('unsupported shape')
    with pytest.raises(vkn.VknLibraryError):   # CPU tensors: no fallback
        vkn.ops.panoptic_joint(torch.zeros(1, 15, 5), torch.zeros(1, 15, 8, 16), 12, 2, 12, 0.25, 0.6, (64, 128), (64, 128), (64, 128))
    with pytest.raises(vkn.VknLibraryError):
        vkn.ops.kernel_init(torch.zeros(1, 64, 8, 16), None, torch.zeros(12, 64, 1, 1))


def test_lsap_matches_scipy(vkn):
    """libvkn's host LSAP (shortest augmenting path, scipy's scan order and tie rule) == scipy.optimize.linear_sum_assignment,
    including rectangular inputs in both orientations and integer matrices full of ties."""
    from scipy.optimize import linear_sum_assignment
    rng = np.random.default_rng(7)
    for trial in range(200):
        nr, nc = int(rng.integers(1, 48)), int(rng.integers(1, 48))
        c = (rng.integers(0, 4, size=(nr, nc)) if trial % 3 == 0 else rng.standard_normal((nr, nc))).astype(np.float32)
        r, cc = vkn.ops.lsap(c)
        r0, c0 = linear_sum_assignment(c)
        assert np.array_equal(r, r0) and np.array_equal(cc, c0)
    g = np.load(os.path.join(GOLDEN, 'assign_cfg.npz'))     # the reference's own cost matrix -> the reference's assignment
    r, cc = vkn.ops.lsap(g['cost'])
    inds = np.zeros(g['cost'].shape[0], dtype=np.int64)
    inds[r] = cc + 1
    assert np.array_equal(inds, g['gt_inds'])
    with pytest.raises(vkn.VknError):
        vkn.ops.lsap(np.full((3, 3), np.nan, dtype=np.float32))


def test_mask_hungarian_assigner_surface(vkn):
    A = vkn.MaskHungarianAssigner
    a = A(cls_cost=dict(type='FocalLossCost', weight=2.0), dice_cost=dict(type='DiceCost', weight=4.0, pred_act=True),
          mask_cost=dict(type='MaskCost', weight=1.0, pred_act=True))
    assert a.cls['weight'] == 2.0 and a.dice['eps'] == 1e-3
    with pytest.raises(NotImplementedError):
        A()                                                # the class defaults name costs no shipped config uses
    with pytest.raises(NotImplementedError):
        A(cls_cost=dict(type='FocalLossCost'), dice_cost=dict(type='DiceCost', pred_act=False), mask_cost=dict(type='MaskCost', pred_act=True))
    with pytest.raises(vkn.VknLibraryError):               # CPU tensors: no fallback
        a.assign(torch.zeros(4, 8, 8), torch.zeros(4, 2), torch.zeros(2, 8, 8), torch.zeros(2, dtype=torch.long))
    r = a.assign(torch.zeros(4, 8, 8), torch.zeros(4, 2), torch.zeros(0, 8, 8), torch.zeros(0, dtype=torch.long))
    assert r.num_gts == 0 and (r.gt_inds == 0).all()


def test_conv_kernel_head_training_surface(vkn):
    """The shipped rpn_head dict (losses + train_cfg.rpn) builds: losses / assigner / sampler as the reference (knet/det/kernel_head.py:
    90-120), no extra state-dict keys, the reference's loss / target helpers are present."""
    g = dict(np.load(os.path.join(GOLDEN, 'init_keys.npz'), allow_pickle=False))
    head = vkn.build_head(dict(
        type='ConvKernelHead', num_classes=19, num_thing_classes=2, num_stuff_classes=17, cat_stuff_mask=True, conv_kernel_size=1,
        feat_downsample_stride=2, feat_refine_stride=1, feat_refine=False, use_binary=True, num_loc_convs=1, num_seg_convs=1,
        conv_normal_init=True, localization_fpn=None, num_proposals=100, proposal_feats_with_obj=True, xavier_init_kernel=False,
        kernel_init_std=1, loss_rank=dict(type='CrossEntropyLoss', use_sigmoid=False, loss_weight=0.1),
        loss_seg=dict(type='FocalLoss', use_sigmoid=True, gamma=2.0, alpha=0.25, loss_weight=1.0),
        loss_mask=dict(type='CrossEntropyLoss', use_sigmoid=True, loss_weight=1.0), loss_dice=dict(type='DiceLoss', loss_weight=4.0),
        train_cfg=dict(assigner=dict(type='MaskHungarianAssigner', cls_cost=dict(type='FocalLossCost', weight=2.0),
                                     dice_cost=dict(type='DiceCost', weight=4.0, pred_act=True),
                                     mask_cost=dict(type='MaskCost', weight=1.0, pred_act=True)),
                       sampler=dict(type='MaskPseudoSampler'), pos_weight=1)))
    assert sorted(head.state_dict()) == list(g['kitti_keys'])
    assert type(head.assigner).__name__ == 'MaskHungarianAssigner' and type(head.sampler).__name__ == 'MaskPseudoSampler'
    assert head.loss_seg.use_sigmoid and head.loss_cls is None and head.loss_rank is not None
    for m in ('forward_train', 'loss', 'get_targets', '_get_target_single', 'simple_test_rpn'):
        assert callable(getattr(head, m))


def test_quasi_dense_tracker_surface(vkn):
    """Registry name / ctor kwargs of the reference's tracker; no CPU path (the association runs in vkn_qd_tracker_match_f32)."""
    cfg = dict(type='QuasiDenseEmbedTracker', init_score_thr=0.35, obj_score_thr=0.3, match_score_thr=0.5, memo_tracklet_frames=5,
               memo_backdrop_frames=1, memo_momentum=0.8, nms_conf_thr=0.5, nms_backdrop_iou_thr=0.3, nms_class_iou_thr=0.7,
               with_cats=True, match_metric='bisoftmax')     # configs/det/video_knet_vipseg/..._joint_train_8e.py: tracker=dict(...)
    trk = vkn.build_tracker(cfg)
    assert trk.empty and trk.num_tracklets == 0 and trk.tracklets == {} and trk.backdrops == []
    with pytest.raises(vkn.VknLibraryError):
        trk.match(torch.zeros(3, 5), torch.zeros(3, dtype=torch.long), torch.zeros(3, 16), 0)
    with pytest.raises(AssertionError):
        vkn.build_tracker(dict(cfg, match_metric='euclid'))
    import ctypes
    L = vkn._lib.lib()
    c = trk._make_cfg(64)
    assert L.vkn_qd_tracker_state_bytes(ctypes.byref(c)) > 0 and L.vkn_qd_tracker_workspace_bytes(ctypes.byref(c)) > 0
    c.max_dets = 1024                                        # outside the envelope: size queries answer 0, calls VKN_E_SHAPE
    assert L.vkn_qd_tracker_state_bytes(ctypes.byref(c)) == 0


def test_joint_merge_helpers_follow_the_rule(vkn):
    """`merge_stuff_thing_stuff_joint` (image and video signatures) and the VIS `merge_stuff_thing`: the vectorised merge against a
    plain loop over the masks in descending score order (the rule of knet/det/kernel_iter_head.py:467-524), on soft masks with
    ties in neither scores nor products; CPU tensors (the helpers are host-side torch)."""
    from types import SimpleNamespace
    g = torch.Generator().manual_seed(5)
    T, Sn, H, W = 3, 4, 12, 18
    head = vkn.build_head(_cfg(True, n_thing=T, n_stuff=Sn, ncls=T + Sn, nprop=6, C=32, heads=4, ffn=64))
    Kt, Ks = 6, Sn
    # every mask owns a column stripe (p = 0.9 there, noise elsewhere); masks 1 and 2 share a stripe (the weaker one keeps too little
    # of its area and is dropped), thing 4 scores below the instance threshold
    base = torch.rand(Kt + Ks, H, W, generator=g) * 0.2
    stripe = [0, 1, 1, 2, 3, 4, 5, 6, 7, 8]
    for k, c in enumerate(stripe):
        base[k, :, 2 * c:2 * c + 2] = 0.9 + 0.05 * torch.rand(H, 2, generator=g)
    tm, sm = base[:Kt], base[Kt:]
    tl, sl = torch.randint(0, T, (Kt,), generator=g), torch.arange(Ks) + T
    ts = torch.tensor([0.91, 0.83, 0.62, 0.77, 0.12, 0.55])
    ss = torch.tensor([0.71, 0.64, 0.93, 0.48])
    cfgm = SimpleNamespace(instance_score_thr=0.3, overlap_thr=0.5, iou_thr=0.5, stuff_max_area=4)
    masks, labels, scores = torch.cat([tm, sm]), torch.cat([tl, sl]), torch.cat([ts, ss])
    win = (scores.view(-1, 1, 1) * masks).argmax(0)
    want, info, things, sid = torch.zeros(H, W, dtype=torch.int32), [], [], 0
    for k in torch.argsort(-scores).tolist():
        thing = int(labels[k]) < T
        if thing and float(scores[k]) < cfgm.instance_score_thr:
            continue
        a, o = int((win == k).sum()), int((masks[k] >= 0.5).sum())
        if a == 0 or o == 0 or a / o < cfgm.overlap_thr:
            continue
        sid += 1
        want[win == k] = sid
        info.append((sid, thing, int(labels[k]) if thing else int(labels[k]) - T + 1, k if thing else a))
        if thing:
            things.append(k)
    assert sid >= 3
    pan, segs = super(type(head), head).merge_stuff_thing_stuff_joint(tm, tl, ts, sm, sl, ss, cfgm)
    assert np.array_equal(pan, want.numpy()) and pan.dtype == np.int32
    assert [(s_['id'], s_['isthing'], s_['category_id'], s_['instance_id'] if s_['isthing'] else s_['area']) for s_ in segs] == info
    obj_t, obj_s = torch.randn(Kt, 8, generator=g), torch.randn(Ks, 8, generator=g)
    (pan2, segs2), feats = head.merge_stuff_thing_stuff_joint(tm, tl, ts, sm, sl, ss, cfgm, thing_obj=obj_t, stuff_obj=obj_s)
    assert np.array_equal(pan2, pan) and segs2 == segs and torch.equal(feats, torch.cat([obj_t, obj_s])[things])
    assert [torch.equal(u, v) for u, v in zip(head.split_thing_stuff(masks, torch.cat([tl, sl]), scores),
                                              (masks[:6], labels[:6], scores[:6], masks[6:], labels[6:] - T + 1, scores[6:]))] == [True] * 6
    vis = vkn.HEADS.get('KernelIterHeadVideo')
    code = vis.merge_stuff_thing(head, masks, labels, scores, cfgm)
    exp = np.full((H, W), head.num_classes, dtype=np.int64)
    order = [k for k in torch.argsort(-scores).tolist()]
    seg_of = {}
    for k in order:
        v = int(want[win == k].max()) if bool((win == k).any()) else 0
        if v > 0:
            seg_of[k] = v
    for k, v in seg_of.items():
        exp[(win == k).numpy()] = int(labels[k]) + (v - 1) * 1000
    assert np.array_equal(code, exp) and code.dtype == np.int64


def test_update_head_single_image_targets_equal_the_batch_builder(vkn):
    """`KernelUpdateHead._get_target_single` (the reference's per-image argument list) and `get_targets(concat=False)` are the batch
    builder on one image: same tensors as slicing the concatenated result."""
    from types import SimpleNamespace
    g = torch.Generator().manual_seed(11)
    head = vkn.build_head(_cfg(False, n_thing=3, n_stuff=2, ncls=5, nprop=7, C=32, heads=4, ffn=64)).mask_head[0]
    N, H, W = 7, 6, 10
    res, segs, clss = [], [], []
    for i in range(2):
        k = 3 - i
        pos = torch.sort(torch.randperm(N, generator=g)[:k])[0]
        res.append(SimpleNamespace(pos_inds=pos, pos_gt_masks=(torch.rand(k, H, W, generator=g) > 0.5).float(),
                                   pos_gt_labels=torch.randint(0, 3, (k,), generator=g), num_pos=k, num_neg=N - k,
                                   device=torch.device('cpu'), mask_dtype=torch.float32, mask_shape=(H, W)))
        clss.append(torch.tensor([3, 4][:2 - i]))
        segs.append((torch.rand(2 - i, H, W, generator=g) > 0.5).float())
    cfg = dict(pos_weight=1)
    whole = head.get_targets(res, None, None, cfg, True, gt_sem_seg=segs, gt_sem_cls=clss)
    lists = head.get_targets(res, None, None, cfg, False, gt_sem_seg=segs, gt_sem_cls=clss)
    Ns = N + 2
    for i in range(2):
        for a, b in zip(whole, (t[i] for t in lists)):
            assert torch.equal(a[i * Ns:(i + 1) * Ns], b)
        r = res[i]
        one = head._get_target_single(r.pos_inds, None, torch.zeros(r.num_pos, H, W), torch.zeros(r.num_neg, H, W), r.pos_gt_masks,
                                      r.pos_gt_labels, segs[i], clss[i], cfg)
        for a, b in zip(one, (t[i] for t in lists)):
            assert torch.equal(a, b)



def test_release_sources_hold_no_debug_only_kernels():
    """The product sources define ONE kernel per job: everything rejected or superseded lives in tools/experiments/*.inc and enters
    only the -DVKN_DEBUG build through #include.  No `__global__` definition may sit inside an `#ifdef VKN_DEBUG` block of csrc/, and
    the release build reads no environment variable (vkn_dbg_env is a constant there)."""
    import glob
    csrc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'video-k-net_amd', 'csrc')
    files = sorted(glob.glob(os.path.join(csrc, '*.hip')) + glob.glob(os.path.join(csrc, '*.h')))
    assert len(files) >= 12
    for f in files:
        stack = []
        for i, line in enumerate(open(f), 1):
            t = line.strip()
            if t.startswith('#if'):
                stack.append('VKN_DEBUG' in t and not t.startswith('#ifndef'))
            elif t.startswith('#else') and stack:
                stack[-1] = False
            elif t.startswith('#endif') and stack:
                stack.pop()
            elif any(stack):
                assert '__global__' not in t, f'{os.path.basename(f)}:{i}: a kernel inside #ifdef VKN_DEBUG'
            else:
                assert 'getenv' not in t or 'VKN_DEBUG' in t or t.startswith('//'), f'{os.path.basename(f)}:{i}: environment read in the release build'


def test_chain_graph_gradient_delivery_survives_inplace_zero_and_accumulation(vkn):
    """`_ChainGraphRunner._deliver` hands the captured backward graph's STATIC gradient buffers over as `p.grad`.  With
    `zero_grad(set_to_none=False)` or a second micro-batch (no zeroing at all) that buffer is still `p.grad` when the next replay
    overwrites it — the replay is simulated here by writing into the buffers, which is all a graph replay does to them."""
    from importlib import import_module
    kuh = import_module('video_k_net_amd.kernel_update_head')
    p = torch.nn.Parameter(torch.zeros(3))
    buf = torch.zeros(3)
    r = kuh._ChainGraphRunner.__new__(kuh._ChainGraphRunner)
    r.used, r.head = [(p, buf)], object()

    def backward(vals):
        r._unalias()                      # what _ChainGraphFn.backward does around the replay
        buf.copy_(torch.tensor(vals))     # "replay"
        r._deliver()

    backward([1., 2., 3.])
    assert p.grad.tolist() == [1., 2., 3.]
    p.grad.zero_()                        # optimizer.zero_grad(set_to_none=False)
    backward([1., 2., 3.])
    assert p.grad.tolist() == [1., 2., 3.]            # was [2, 4, 6]: the buffer added to itself
    backward([10., 10., 10.])             # gradient accumulation: no zeroing between two backwards
    assert p.grad.tolist() == [11., 12., 13.]         # was [20, 20, 20]: the first micro-batch lost
    p.grad = None                         # the usual loop
    backward([5., 6., 7.])
    assert p.grad.tolist() == [5., 6., 7.] and p.grad.data_ptr() == buf.data_ptr()   # fast path still hands the buffer itself over


def test_python_flag_constants_match_the_header(vkn):
    """`ops.FLAG_*` / `ops.PHASE_*` are typed by hand: they must equal include/vkn.h's VKN_FLAG_* values (and no two flags may share a bit)."""
    import re
    text = open(os.path.join(ROOT, 'include', 'vkn.h')).read()
    hdr = {m.group(1): int(m.group(2)) for m in re.finditer(r'#define VKN_FLAG_(\w+) (\d+)u', text)}
    assert len(hdr) >= 12 and len(set(hdr.values())) == len(hdr)
    assert all(v & (v - 1) == 0 for v in hdr.values()), 'every flag is one bit'
    py = dict(REF_KERNELS=vkn.ops.FLAG_REF_KERNELS, EXACT_GEMM=vkn.ops.FLAG_EXACT_GEMM, LOGITS_HANDOFF=vkn.ops.FLAG_LOGITS_HANDOFF,
              BITS_HANDOFF=vkn.ops.FLAG_BITS_HANDOFF, SERIAL_LINK=vkn.ops.FLAG_SERIAL_LINK, X_F16=vkn.ops.FLAG_X_F16,
              X_BF16=vkn.ops.FLAG_X_BF16, CHAIN_LAUNCHES=vkn.ops.FLAG_CHAIN_LAUNCHES, CHAIN_PERSISTENT=vkn.ops.FLAG_CHAIN_PERSISTENT, CHAIN_BF16X3=vkn.ops.FLAG_CHAIN_BF16X3,
              PHASE_A=vkn.ops.PHASE_A, PHASE_B=vkn.ops.PHASE_B, PHASE_C=vkn.ops.PHASE_C, CLIP_LINK=8)
    for k, v in py.items():
        assert hdr[k] == v, (k, hdr[k], v)
    assert int(re.search(r'#define VKN_E_RANGE \((-\d+)\)', text).group(1)) == -6
    assert vkn._lib.lib().vkn_strerror(-6).decode().startswith('feature map outside')


@pytest.mark.parametrize('kind', ['image', 'video_ffn', 'video_update', 'video_update_obj'])
def test_training_chain_enumerates_every_linear_parameter_of_a_stage(vkn, kind):
    """`chain_train.chain_linears` lists the Linear layers whose tile images are built (and whose weight gradients are batched) per chain
    forward; a layer missing from it would still train — through the per-layer path — but slowly and unnoticed.  Host logic only: the
    owners of the listed (views of) weights and biases are exactly the stage's parameters that are not LayerNorm vectors."""
    import torch.nn as nn
    over = {'video_update': dict(previous_link='update_dynamic_cov', previous_type='update'),
            'video_update_obj': dict(previous_link='link_atten', previous_type='update_obj')}.get(kind)
    cfgd = vkn.configs.roi_head_cfg(kind != 'image', C=64, heads=8, ffn=128, ncls=19, n_thing=8, n_stuff=11, S=1, up=2, nprop=20,
                                    train_cfg=vkn.configs.rcnn_train_cfg(1), mask_over=over)
    stage = vkn.build_head(cfgd).mask_head[0]
    ct = vkn.chain_train
    linears = ct.chain_linears(stage, kind != 'image')
    owners = {id(p) for p in ct._owners(linears, list(stage.parameters()))}
    norm_params = {id(p) for m in stage.modules() if isinstance(m, nn.LayerNorm) for p in m.parameters()}
    expected = {id(p) for p in stage.parameters()} - norm_params
    names = {id(p): n for n, p in stage.named_parameters()}
    assert owners == expected, sorted(names[i] for i in owners ^ expected)


def test_public_header_is_plain_c(tmp_path):
    """include/vkn.h is the drop-in boundary: a C99 translation unit that only includes it (and names the structs the bindings mirror)
    must compile — no C++-isms, no torch / HIP types in the signatures."""
    import shutil
    import subprocess
    gcc = shutil.which('gcc')
    if gcc is None:
        pytest.skip('no gcc in this environment')
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = tmp_path / 'hdr.c'
    src.write_text('#include "include/vkn.h"\n'
                   'int main(void) { VknDims d; VknStageWeights w; VknSplitItem a; VknDwItem b; VknUpdatorNorms c; VknUpdatorNormGrads g;\n'
                   '  (void)d; (void)w; (void)a; (void)b; (void)c; (void)g; return vkn_version() == 0; }\n')
    r = subprocess.run([gcc, '-std=c99', '-Wall', '-Wextra', '-Werror', '-fsyntax-only', '-I', root, str(src)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


def test_device_assign_result_is_lazy_and_equal_to_the_eager_fields(vkn):
    """`DeviceAssignResult` (what `assign_batch` returns on the device path): `labels` and `device_pos_inds` are built on first access
    from the LSAP's (row, col) pairs and equal the fields the eager construction produced (mask_hungarian_assigner.py, reference
    :262-274); both can be overwritten like the attributes of mmdet's AssignResult."""
    from importlib import import_module
    mha = import_module('video_k_net_amd.mask_hungarian_assigner')
    N, G = 12, 4
    rows, cols = torch.tensor([1, 4, 7, 10], dtype=torch.int32), torch.tensor([2, 0, 3, 1], dtype=torch.int32)
    gt_labels = torch.tensor([5, 6, 7, 8])
    gt_inds = torch.zeros(N, dtype=torch.int64)
    gt_inds[rows.long()] = cols.long() + 1
    r = mha.DeviceAssignResult(G, gt_inds, (rows, cols), gt_labels, None)
    assert r._labels is None and r._pos is None and r.num_gts == G and r.max_overlaps is None
    want = torch.full((N,), -1, dtype=torch.long)
    want[rows.long()] = gt_labels[cols.long()]
    assert torch.equal(r.labels, want) and r.labels is r.labels
    assert torch.equal(r.device_pos_inds, rows.long()) and r.device_pos_inds.dtype == torch.int64
    s = import_module('video_k_net_amd.mask_pseudo_sampler').MaskPseudoSampler().sample(r, torch.zeros(N, 2, 2), torch.zeros(G, 2, 2))
    assert torch.equal(s.pos_inds, rows.long()) and torch.equal(s.pos_gt_labels, gt_labels[cols.long()]) and s.num_neg == N - G
    assert torch.equal(s.pos_assigned_gt_inds, cols.long())
    r.labels = want + 1
    assert torch.equal(r.labels, want + 1)


def test_fused_training_tail_declines_what_it_does_not_cover(vkn):
    """`TailStep.begin` returns None — the op-by-op path then runs — for CPU tensors, foreign assigners / samplers / loss objects, empty
    or oversized ground truth; it never raises."""
    from importlib import import_module
    tt = import_module('video_k_net_amd.train_tail')
    head = vkn.build_head(vkn.configs.roi_head_cfg(False, C=64, heads=8, ffn=128, ncls=5, n_thing=2, n_stuff=3, S=2, up=2, nprop=12,
                                                   train_cfg=vkn.configs.rcnn_train_cfg(2)))
    gm, gl = [torch.zeros(3, 16, 32)], [torch.zeros(3, dtype=torch.long)]
    assert tt.TailStep.begin(head, torch.device('cpu'), gm, gl, None, None) is None
    dev = torch.device('cuda', 0)          # (never touched: every check below fails before a tensor would be moved)
    assert tt.TailStep.begin(head, dev, gm, gl, None, None) is None                                  # ground truth on another device
    assert tt.TailStep.begin(head, dev, [], [], None, None) is None
    head.fused_tail = False
    assert tt.TailStep.begin(head, dev, gm, gl, None, None) is None
    head.fused_tail = True
    head.mask_assigner[0].lsap = 'host'
    assert tt.TailStep.begin(head, dev, gm, gl, None, None) is None
    head.mask_assigner[0].lsap = 'device'
    head.mask_head[1].fused_mask_losses = False
    assert tt.TailStep.begin(head, dev, gm, gl, None, None) is None

"""CPU: the oracle (oracle/knet_oracle.py) against golden vectors captured from the reference's own Python
(oracle/gen_golden.py).  This is what pins the oracle (the reference has no tests of its own)."""
import os

import numpy as np
import pytest
import torch

from helpers import GOLDEN, cfg_of, load_golden, maxabs, run_oracle
from oracle.knet_oracle import binarize, head_param_shapes

TOL = 2e-5   # same ATen ops, same machine class: summation-order noise only (|logit| up to ~60)


@pytest.mark.parametrize('name', ['det_tiny', 'det_odd', 'det_cfg', 'video_tiny', 'video_cfg'])
def test_oracle_matches_reference_golden(name):
    g, case = load_golden(name)
    traces = []
    obj, cls, masks, scaled, track = run_oracle(case, traces=traces)
    assert maxabs(obj, g['object_feats']) < TOL
    assert maxabs(cls, g['cls_score']) < TOL
    assert maxabs(masks, g['mask_preds']) < 5 * TOL
    for s, tr in enumerate(traces):
        assert maxabs(tr['cls_score'], g[f's{s}_cls']) < TOL
        assert maxabs(tr['obj_feat'], g[f's{s}_obj']) < TOL
        if f's{s}_mask' in g:
            assert maxabs(tr['new_mask_preds'], g[f's{s}_mask']) < 5 * TOL
    if 'scaled_mask_preds' in g:
        assert maxabs(scaled, g['scaled_mask_preds']) < 5 * TOL
    else:
        rs = scaled.double().sum(dim=(-1, -2)).numpy()
        assert np.max(np.abs(rs - g['scaled_rowsum']) / (1.0 + np.abs(g['scaled_rowsum']))) < 1e-4
    if case['video']:
        assert track is not None and maxabs(track, g['track']) < TOL
    else:
        assert track is None


@pytest.mark.parametrize('name', ['video_upd_tiny', 'video_updffn_tiny', 'video_upd_cfg', 'video_latt_upd_tiny', 'video_updobj_tiny',
                                  'video_latt_updobj_tiny'])
def test_oracle_update_link_heads_match_reference_golden(name):
    """previous_link='update_dynamic_cov' / previous_type='update' (knet/video/kernel_update_head.py:324-348, 417-445), incl. the
    frame-by-frame walk of a video where frame t's last stage is linked to frame t-1's final kernels."""
    from helpers import make_case
    from oracle.knet_oracle import iter_head_mask_preds
    g, case = load_golden(name)
    shapes = head_param_shapes(cfg_of(case))
    assert sorted(shapes) == list(g['keys']) and [str(tuple(shapes[k])) for k in sorted(shapes)] == list(g['shapes'])
    obj, cls, masks, scaled, track = run_oracle(case)
    assert maxabs(obj, g['object_feats']) < TOL and maxabs(cls, g['cls_score']) < TOL
    assert maxabs(masks, g['mask_preds']) < 5 * TOL and maxabs(track, g['track']) < TOL
    cfg, sd, x, pf, mp, _ = make_case(case)
    memo = None
    with torch.no_grad():
        for t in range(case['B']):
            o, c, m, _, tr = iter_head_mask_preds(sd, x[t:t + 1], pf[t:t + 1], mp[t:t + 1], cfg, previous_obj_feats=memo)
            memo = o
            assert maxabs(o, g[f'clip_obj{t}']) < 2 * TOL and maxabs(c, g[f'clip_cls{t}']) < TOL
            if f'clip_mask{t}' in g:
                assert maxabs(m, g[f'clip_mask{t}']) < 5 * TOL
            if f'clip_track{t}' in g:
                assert maxabs(tr, g[f'clip_track{t}']) < 2 * TOL


def test_oracle_matches_reference_golden_cfg1_size():
    g, case = load_golden('det_cfg_big')
    obj, cls, masks, scaled, _ = run_oracle(case)
    assert maxabs(obj, g['object_feats']) < TOL
    assert maxabs(cls, g['cls_score']) < TOL
    flat = masks.reshape(-1)
    assert maxabs(flat[torch.from_numpy(g['sample_idx'])], g['sample_val']) < 1e-4
    rs = masks.double().sum(dim=(-1, -2)).numpy()
    assert np.max(np.abs(rs - g['mask_rowsum'])) < 1e-4 * np.max(g['mask_rowabs'])
    bits = np.packbits(flat.numpy() > 0)
    valid = g['sign_valid']
    assert np.all((bits ^ g['sign_bits']) & valid == 0), 'sign of a logit with |logit|>2e-3 differs from the reference'


def test_state_dict_keys_match_reference():
    for name in ('det_cfg', 'video_cfg'):
        g, case = load_golden(name)
        shapes = head_param_shapes(cfg_of(case))
        assert sorted(shapes) == list(g['keys'])
        assert [str(tuple(shapes[k])) for k in sorted(shapes)] == list(g['shapes'])
    g, case = load_golden('det_cfg')
    assert len([k for k in g['keys'] if k.startswith('mask_head.0.')]) == 44
    n = sum(int(np.prod(eval(s))) for k, s in zip(g['keys'], g['shapes']) if k.startswith('mask_head.0.'))
    assert n == 2046739


def test_threshold_kat():
    g = np.load(os.path.join(GOLDEN, 'thr_kat.npz'))
    z = torch.from_numpy(g['z'])
    assert np.array_equal(binarize(z, 0.5).numpy() > 0, g['bit'])
    assert float(g['flip']) == pytest.approx(8.94069742685133e-08, rel=0, abs=0)
    # not "z > 0": strictly positive logits below the flip point stay OFF
    assert not bool(binarize(torch.tensor([5e-8]), 0.5)[0]) and bool(binarize(torch.tensor([1e-7]), 0.5)[0])


def test_oracle_fp64_headroom():
    """fp32 oracle vs the same code in fp64: documents the tolerance budget."""
    _, case = load_golden('det_cfg')
    o32 = run_oracle(case, torch.float32)
    o64 = run_oracle(case, torch.float64)
    assert maxabs(o32[2], o64[2]) < 1e-3


@pytest.mark.parametrize('name', ['init_tiny', 'init_odd', 'init_cfg'])
def test_kernel_init_oracle_matches_reference_golden(name):
    """oracle.kernel_init == the reference's ConvKernelHead.simple_test_rpn (behind a pass-through neck)."""
    from helpers import load_init_golden, make_init_case
    from oracle.knet_oracle import kernel_init
    g, case = load_init_golden(name)
    loc, sem, iw, sw, sb = make_init_case(case)
    with torch.no_grad():
        prop, xf, masks, seg = kernel_init(iw, loc, sem, sw, sb, case['n_thing'], bool(case['cat']))
    assert tuple(prop.shape) == g['proposal_feats'].shape and tuple(masks.shape) == g['mask_preds'].shape
    assert maxabs(masks, g['mask_preds']) < 1e-5
    assert maxabs(prop, g['proposal_feats']) < 2e-4     # sums of ~P/2 unit-variance features
    if 'x_feats' in g:
        assert maxabs(xf, g['x_feats']) == 0.0
    else:
        assert maxabs(xf.double().sum(dim=(-1, -2)), g['x_feats_rowsum']) < 1e-9
    if seg is not None:
        assert maxabs(seg, g['seg_preds']) < 1e-5


@pytest.mark.parametrize('name', ['pan_tiny', 'pan_ident', 'pan_cfg', 'pan_kitti', 'pan_vipseg'])
def test_panoptic_oracle_matches_reference_golden(name):
    """oracle.panoptic_joint == the reference's KernelIterHead.get_panoptic (merge_joint=True): the integer panoptic map and the
    segments_info list, bit for bit (same ATen ops on the same machine class)."""
    from helpers import load_pan_golden, pan_info_rows, run_pan_oracle
    g, case = load_pan_golden(name)
    for b in range(case['B']):
        r = run_pan_oracle(case, b)
        assert np.array_equal(r['panoptic_seg'].numpy(), g['panoptic_seg'][b])
        rows, want = pan_info_rows(r['segments_info']), g[f'info{b}']
        assert rows.shape == want.shape and int(g['nseg'][b]) == len(r['segments_info']) > 0
        assert np.array_equal(np.nan_to_num(rows, nan=-7.0), np.nan_to_num(want, nan=-7.0))


@pytest.mark.parametrize('name', ['assign_tiny', 'assign_cfg', 'assign_odd'])
def test_assign_oracle_matches_reference_golden(name):
    """oracle.assign_costs / hungarian_assign == the reference's MaskHungarianAssigner with the shipped costs: the cost matrix
    to fp32 rounding, the integer assignment bit-exact."""
    from helpers import load_assign_golden, make_assign_case
    from oracle.knet_oracle import assign_costs, hungarian_assign
    g, case = load_assign_golden(name)
    logits, cls, gt, labels = make_assign_case(case)
    with torch.no_grad():
        cost = assign_costs(logits, cls, gt, labels)
        gt_inds, lab = hungarian_assign(cost, labels)
    assert maxabs(cost, g['cost']) < 1e-6
    assert np.array_equal(gt_inds.numpy(), g['gt_inds']) and np.array_equal(lab.numpy(), g['labels'])


def test_tracker_oracle_ids_bit_exact_vs_reference():
    """oracle/tracker_oracle.py over the four synthetic videos: surviving detections, labels and ids per frame equal the
    reference's own tracker class (oracle/gen_golden_tracker.py) for the three match metrics."""
    from oracle import synth
    from oracle.tracker_oracle import TrackerOracle
    g = dict(np.load(os.path.join(GOLDEN, 'qd_tracker.npz'), allow_pickle=False))
    cfg = dict(init_score_thr=0.35, obj_score_thr=0.3, match_score_thr=0.5, memo_tracklet_frames=5, memo_backdrop_frames=1,
               memo_momentum=0.8, nms_conf_thr=0.5, nms_backdrop_iou_thr=0.3, nms_class_iou_thr=0.7, with_cats=True)
    for name in ('trk_a', 'trk_b', 'trk_c', 'trk_d'):
        T, n_obj, emb, n_cls, seed = (int(v) for v in g[name + '_case'])
        trk = TrackerOracle(**cfg, match_metric=str(g[name + '_metric']))
        for t, (bb, lab, em, _) in enumerate(synth.tracker_sequence(T, n_obj, emb, n_cls, seed)):
            b, l_, ids = trk.step(torch.from_numpy(bb), torch.from_numpy(lab), torch.from_numpy(em), t)
            assert np.array_equal(ids.numpy(), g[f'{name}_ids{t}']), (name, t)
            assert np.array_equal(l_.numpy(), g[f'{name}_labels{t}']) and np.array_equal(b.numpy(), g[f'{name}_bboxes{t}'])


@pytest.mark.parametrize('name', ['vis_attn_tiny', 'vis_attnpos_tiny', 'vis_attnpos_cfg'])
def test_query_merge_oracle_matches_reference_golden(name):
    """The clip-level attention query merge (query_merge_method 'attention' / 'attention_pos'): the oracle's restatement on the
    reference roi head's object features reproduces the reference tracker's own `_query_fusion` output."""
    import ast
    from oracle import synth
    from oracle.knet_oracle import query_merge
    g = dict(np.load(os.path.join(GOLDEN, name + '.npz'), allow_pickle=False))
    C, N, bs, nf, seed = (int(g['case'][i]) for i in (0, 4, 9, 10, 11))
    shapes = {str(k): ast.literal_eval(str(v)) for k, v in zip(g['trk_keys'], g['trk_shapes'])}
    sd = {k: torch.from_numpy(v) for k, v in synth.state_dict_like(shapes, seed + 1).items()}
    keys = torch.from_numpy(g['roi_obj_feats']).reshape(bs, nf * N, C)
    query = sd['init_query.weight'].expand(bs, N, C)
    pos = sd['query_pos.weight'] if str(g['merge']) == 'attention_pos' else None
    assert ('query_pos.weight' in sd) == (pos is not None)
    out = query_merge(sd, '', query, keys, pos)
    ref = g['trk_query_fusion']
    assert tuple(ref.shape) == (bs, N, C, 1, 1)
    assert maxabs(out[..., None, None], ref) < 2e-5

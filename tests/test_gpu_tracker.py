"""GPU (-m gpu): the device-side quasi-dense association (`vkn_qd_tracker_match_f32`, one single-workgroup kernel per frame over a
device-resident memo) against the reference's own tracker (tests/golden/qd_tracker.npz, bit-exact ids / labels / boxes) and, on
videos far larger than the goldens, against the CPU oracle (oracle/tracker_oracle.py, itself pinned to the same goldens)."""
import os

import numpy as np
import pytest
import torch

from helpers import GOLDEN
from oracle import synth
from oracle.tracker_oracle import TrackerOracle, random_video

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
CFG = dict(init_score_thr=0.35, obj_score_thr=0.3, match_score_thr=0.5, memo_tracklet_frames=5, memo_backdrop_frames=1,
           memo_momentum=0.8, nms_conf_thr=0.5, nms_backdrop_iou_thr=0.3, nms_class_iou_thr=0.7, with_cats=True)


def _step(trk, bb, lab, em, t):
    return trk.match(torch.from_numpy(bb).to(DEV), torch.from_numpy(lab).to(DEV), torch.from_numpy(em).to(DEV), t)


@pytest.mark.parametrize('name', ['trk_a', 'trk_b', 'trk_c', 'trk_d'])
def test_tracker_ids_bit_exact_vs_reference(vkn, name):
    g = dict(np.load(os.path.join(GOLDEN, 'qd_tracker.npz'), allow_pickle=False))
    T, n_obj, emb, n_cls, seed = (int(v) for v in g[name + '_case'])
    trk = vkn.build_tracker(dict(CFG, type='QuasiDenseEmbedTracker', match_metric=str(g[name + '_metric'])))
    for t, (bb, lab, em, _) in enumerate(synth.tracker_sequence(T, n_obj, emb, n_cls, seed)):
        b, l_, ids = _step(trk, bb, lab, em, t)
        assert not ids.is_cuda and ids.dtype == torch.int64
        assert np.array_equal(ids.numpy(), g[f'{name}_ids{t}']), (name, t)
        assert np.array_equal(l_.cpu().numpy(), g[f'{name}_labels{t}']) and np.array_equal(b.cpu().numpy(), g[f'{name}_bboxes{t}'])
    assert trk.num_tracklets == int(max(g[f'{name}_ids{t}'].max() for t in range(T))) + 1


@pytest.mark.parametrize('metric', ['bisoftmax', 'softmax', 'cosine'])
@pytest.mark.parametrize('bd_frames', [0, 1, 3])
def test_tracker_vs_oracle_on_dense_videos(vkn, metric, bd_frames):
    """120 objects, 256-d embeddings, 14 frames, tracks expiring after 3 unseen frames: every per-frame decision (survivors, ids),
    and at the end the whole memo (track table in creation order, momentum embeddings, velocities, backdrops)."""
    cfg = dict(CFG, match_metric=metric, memo_backdrop_frames=bd_frames, memo_tracklet_frames=3, init_score_thr=0.5, obj_score_thr=0.35)
    trk = vkn.build_tracker(dict(cfg, type='QuasiDenseEmbedTracker', max_tracklets=1024))
    ora = TrackerOracle(**cfg)
    for t, (bb, lab, em) in enumerate(random_video(14, 120, 256, 4, 7 + bd_frames)):
        b, l_, ids = _step(trk, bb, lab, em, t)
        rb, rl, rids = ora.step(torch.from_numpy(bb), torch.from_numpy(lab), torch.from_numpy(em), t)
        assert np.array_equal(b.cpu().numpy(), rb.numpy()) and np.array_equal(l_.cpu().numpy(), rl.numpy()), t
        assert np.array_equal(ids.numpy(), rids.numpy()), (t, np.nonzero(ids.numpy() != rids.numpy()))
    tr = trk.tracklets
    assert list(tr) == ora.t_id and trk.num_tracklets == ora.next_id
    for i, tid in enumerate(ora.t_id):
        e = tr[tid]
        assert e['last_frame'] == ora.t_last[i] and e['acc_frame'] == ora.t_acc[i] and e['label'] == int(ora.t_label[i])
        assert torch.equal(e['bbox'], ora.t_box[i])
        assert float((e['embed'] - ora.t_emb[i]).abs().max()) == 0.0, 'momentum embedding: same fp32 operation sequence'
        assert float((e['velocity'] - ora.t_vel[i]).abs().max()) < 1e-6
    bds = trk.backdrops
    assert len(bds) == len(ora.backdrops)
    for a, r in zip(bds, ora.backdrops):
        assert torch.equal(a['bboxes'], r['box']) and torch.equal(a['embeds'], r['emb']) and torch.equal(a['labels'].long(), r['label'])


def test_tracker_padded_api_and_edge_cases(vkn):
    trk = vkn.build_tracker(dict(CFG, type='QuasiDenseEmbedTracker', max_dets=64, max_tracklets=8))
    z = lambda *s, dt=torch.float32: torch.zeros(*s, dtype=dt, device=DEV)  # noqa: E731
    b, l_, ids = trk.match(z(0, 5), z(0, dt=torch.long), z(0, 32), 0)                      # an empty frame
    assert b.shape == (0, 5) and ids.numel() == 0 and trk.empty
    bb, lab, em = random_video(1, 40, 32, 2, 3)[0]
    ob, ol, oi, cnt = trk.match_padded(torch.from_numpy(bb).to(DEV), torch.from_numpy(lab).to(DEV), torch.from_numpy(em).to(DEV), 1)
    assert ob.shape == (64, 5) and cnt.dtype == torch.int32 and oi.is_cuda
    k, status = cnt.cpu().tolist()
    assert 0 < k <= bb.shape[0]
    assert status == 1, 'more births than max_tracklets = 8: reported, not silently dropped'
    with pytest.raises(RuntimeError):
        trk.match(torch.from_numpy(bb).to(DEV), torch.from_numpy(lab).to(DEV), torch.from_numpy(em).to(DEV), 2)
    trk.reset()
    assert trk.empty and trk.num_tracklets == 0
    with pytest.raises(ValueError):
        trk.match(z(65, 5), z(65, dt=torch.long), z(65, 32), 0)

#!/usr/bin/env python3
"""Goldens of the quasi-dense embed tracker AND of the embedding head in front of it from the REFERENCE's own files (build container only).
`knet/video/qdtrack/trackers/quasi_dense_embed_tracker.py` is loaded UNMODIFIED by path — its package `__init__` also imports the
TAO tracker, which needs cv2 / seaborn — with `..builder` loaded the same way; `mmdet.core.bbox_overlaps` is a stand-in."""
import importlib.util
import os
import sys
import types

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = os.environ.get('VKN_REFERENCE', '/root/reference')
sys.path.insert(0, os.path.join(HERE, 'standins'))
sys.path.insert(1, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

from oracle import synth  # noqa: E402


def load_reference_tracker():
    for pkg in ('knet', 'knet.video', 'knet.video.qdtrack', 'knet.video.qdtrack.trackers'):
        m = types.ModuleType(pkg)
        m.__path__ = []
        sys.modules[pkg] = m
    for name, rel in (('knet.video.qdtrack.builder', 'knet/video/qdtrack/builder.py'),
                      ('knet.video.qdtrack.trackers.quasi_dense_embed_tracker', 'knet/video/qdtrack/trackers/quasi_dense_embed_tracker.py')):
        spec = importlib.util.spec_from_file_location(name, os.path.join(REF, rel))
        mod = importlib.util.module_from_spec(spec)
        sys.modules[name] = mod
        spec.loader.exec_module(mod)
    return sys.modules['knet.video.qdtrack.trackers.quasi_dense_embed_tracker'].QuasiDenseEmbedTracker


def load_reference_embed_head():
    """`QuasiDenseMaskEmbedHeadGTMask` + its two losses from the reference's own files (knet/video/track_heads.py:552-718,
    knet/video/qdtrack/losses/{multipos_cross_entropy_loss,l2_loss}.py, knet/video/qdtrack/track/similarity.py), loaded UNMODIFIED by
    path.  Shims are names only (no arithmetic): `build_roi_extractor`, `bbox2roi`, `unitrack.utils.mask` (the RoI-based heads of the
    same file import them), and the `mmdet.models` re-exports of the stand-in loss helpers."""
    import mmdet.core as mcore
    import mmdet.models as mmodels
    import mmdet.models.builder as mbuilder
    import mmdet.models.losses.utils as mutils
    mmodels.LOSSES, mmodels.weight_reduce_loss, mmodels.weighted_loss = mbuilder.LOSSES, mutils.weight_reduce_loss, mutils.weighted_loss
    mbuilder.build_roi_extractor = lambda cfg: None
    mcore.bbox2roi = lambda *a, **k: None
    um = types.ModuleType('unitrack.utils.mask')
    um.mask2box = um.batch_mask2boxlist = um.bboxlist2roi = lambda *a, **k: None
    uu = types.ModuleType('unitrack.utils')
    uu.__path__ = []
    sys.modules.setdefault('unitrack.utils', uu)
    sys.modules['unitrack.utils.mask'] = um
    for pkg in ('knet', 'knet.video', 'knet.video.qdtrack', 'knet.video.qdtrack.track', 'knet.video.qdtrack.losses'):
        if pkg not in sys.modules:
            m = types.ModuleType(pkg)
            m.__path__ = []
            sys.modules[pkg] = m

    def by_path(name, rel):
        spec = importlib.util.spec_from_file_location(name, os.path.join(REF, rel))
        mod = importlib.util.module_from_spec(spec)
        sys.modules[name] = mod
        spec.loader.exec_module(mod)
        return mod
    sim = by_path('knet.video.qdtrack.track.similarity', 'knet/video/qdtrack/track/similarity.py')
    sys.modules['knet.video.qdtrack.track'].cal_similarity = sim.cal_similarity
    by_path('knet.video.qdtrack.losses.multipos_cross_entropy_loss', 'knet/video/qdtrack/losses/multipos_cross_entropy_loss.py')
    by_path('knet.video.qdtrack.losses.l2_loss', 'knet/video/qdtrack/losses/l2_loss.py')
    return by_path('knet.video.track_heads', 'knet/video/track_heads.py').QuasiDenseMaskEmbedHeadGTMask


from oracle.embed_cases import EMBED_CASES, embed_case_inputs  # noqa: E402

CFG = dict(init_score_thr=0.35, obj_score_thr=0.3, match_score_thr=0.5, memo_tracklet_frames=5, memo_backdrop_frames=1,
           memo_momentum=0.8, nms_conf_thr=0.5, nms_backdrop_iou_thr=0.3, nms_class_iou_thr=0.7, with_cats=True,
           match_metric='bisoftmax')      # configs/det/video_knet_vipseg/..._joint_train_8e.py: tracker=dict(...)

if __name__ == '__main__':
    Tracker = load_reference_tracker()
    out = dict()
    for name, (T, n_obj, emb, n_cls, seed, metric) in dict(trk_a=(8, 9, 32, 2, 1, 'bisoftmax'), trk_b=(10, 14, 64, 3, 2, 'bisoftmax'),
                                                           trk_c=(6, 7, 16, 1, 3, 'softmax'), trk_d=(6, 7, 16, 2, 4, 'cosine')).items():
        trk = Tracker(**dict(CFG, match_metric=metric))
        frames = synth.tracker_sequence(T, n_obj, emb, n_cls, seed)
        out[name + '_case'] = np.array([T, n_obj, emb, n_cls, seed], dtype=np.int64)
        out[name + '_metric'] = np.array(metric)
        for t, (bb, lab, em, who) in enumerate(frames):
            b, l_, ids = trk.match(bboxes=torch.from_numpy(bb), labels=torch.from_numpy(lab), track_feats=torch.from_numpy(em), frame_id=t)
            out[f'{name}_bboxes{t}'], out[f'{name}_labels{t}'], out[f'{name}_ids{t}'] = b.numpy(), l_.numpy(), ids.numpy()
        print(name, 'ok  tracklets created:', int(trk.num_tracklets), ' last frame ids:', ids.tolist())
    np.savez_compressed(os.path.join(os.environ.get('VKN_GOLDEN_OUT', os.path.join(ROOT, 'tests', 'golden')), 'qd_tracker.npz'), **out)

    # ---- the embedding head between the update head's tracking kernels and the tracker
    Head = load_reference_embed_head()
    out = dict()
    for name, (cfg, sizes, seed) in EMBED_CASES.items():
        head = Head(**cfg)
        sd, keys, refs, kres, rres, match = embed_case_inputs(cfg, sizes, seed)
        head.load_state_dict(sd, strict=True)
        out[name + '_keys'] = np.array(sorted(head.state_dict()))
        with torch.no_grad():
            ke, re_ = head(torch.cat(keys, 0)), head(torch.cat(refs, 0))        # `_track_forward`: cat over the images, then the head
        dists, cos = head.match(ke, re_, kres, rres)
        targets, weights = head.get_track_targets(match, kres, rres)
        out[name + '_key_embeds'], out[name + '_ref_embeds'] = ke.numpy(), re_.numpy()
        for i in range(2):
            out[f'{name}_dists{i}'] = dists[i].numpy()
            if cos[i] is not None:
                out[f'{name}_cos{i}'] = cos[i].numpy()
            out[f'{name}_targets{i}'], out[f'{name}_weights{i}'] = targets[i].numpy(), weights[i].numpy()
        losses = head.loss([d.clone() for d in dists], [c.clone() if c is not None else None for c in cos],
                           [t.clone() for t in targets], [w.clone() for w in weights])
        for k, v in losses.items():
            out[f'{name}_{k}'] = np.float64(float(v))
        print(name, 'ok', {k: round(float(v), 6) for k, v in losses.items()}, 'positives per image', [int(t.sum()) for t in targets])
    np.savez_compressed(os.path.join(os.environ.get('VKN_GOLDEN_OUT', os.path.join(ROOT, 'tests', 'golden')), 'qd_embed_head.npz'), **out)

#!/usr/bin/env python3
"""Golden of the quasi-dense embed tracker from the REFERENCE's own file (build container only).
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

"""`QuasiDenseEmbedTracker` — the association step of the video models (`tracker=dict(type='QuasiDenseEmbedTracker', ...)`,
knet/video/qdtrack/trackers/quasi_dense_embed_tracker.py:9-207) on the MI355X.

Design (not the reference's): the memo is a fixed-capacity structure-of-arrays table in DEVICE memory and one frame is ONE
single-workgroup HIP kernel (`vkn_qd_tracker_match_f32`, csrc/vkn_tracker.hip): score sort, duplicate suppression, the [n x m]
similarity, the order-dependent greedy assignment, births, momentum update, backdrops and expiry all happen there.  The inputs
(thing boxes from `vkn_panoptic_joint_f32`, tracking embeddings from the head) are already on the device and stay there; this
class only owns the state buffer and mirrors the reference's call surface:

    tracker.match(bboxes [n,5], labels [n], track_feats [n,E], frame_id) -> (bboxes [k,5], labels [k], ids [k])

`match` reads back the k ids + the count in one small copy (the reference returns `ids` as a host tensor too);
`match_padded` returns the padded device buffers and the device-side count with no synchronisation at all.
There is no CPU path: CPU tensors raise `VknLibraryError`.
"""
import ctypes

import torch

from . import _lib, ops
from .registry import Registry

TRACKERS = Registry('tracker')
_METRICS = {'bisoftmax': 0, 'softmax': 1, 'cosine': 2}


def build_tracker(cfg):
    return TRACKERS.build(cfg)


@TRACKERS.register_module()
class QuasiDenseEmbedTracker:
    """Ctor kwargs of the reference (:11-38) + two capacities of the device memo: `max_dets` detections per frame (<= 256) and
    `max_tracklets` live tracks (a birth beyond it is dropped and reported through `status`)."""

    def __init__(self, init_score_thr=0.8, obj_score_thr=0.5, match_score_thr=0.5, memo_tracklet_frames=10, memo_backdrop_frames=1,
                 memo_momentum=0.8, nms_conf_thr=0.5, nms_backdrop_iou_thr=0.3, nms_class_iou_thr=0.7, with_cats=True,
                 match_metric='bisoftmax', max_dets=256, max_tracklets=1024):
        if not 0 <= memo_momentum <= 1.0 or memo_tracklet_frames < 0 or memo_backdrop_frames < 0:
            raise AssertionError('memo_momentum in [0, 1], memo_*_frames >= 0')
        if match_metric not in _METRICS:
            raise AssertionError(f'match_metric must be one of {sorted(_METRICS)}')
        self.init_score_thr, self.obj_score_thr, self.match_score_thr = init_score_thr, obj_score_thr, match_score_thr
        self.memo_tracklet_frames, self.memo_backdrop_frames, self.memo_momentum = memo_tracklet_frames, memo_backdrop_frames, memo_momentum
        self.nms_conf_thr, self.nms_backdrop_iou_thr, self.nms_class_iou_thr = nms_conf_thr, nms_backdrop_iou_thr, nms_class_iou_thr
        self.with_cats, self.match_metric = with_cats, match_metric
        self.max_dets, self.max_tracklets = int(max_dets), int(max_tracklets)
        self._cfg = self._state = self._ws = self._layout = None
        self._device = None

    # ------------------------------------------------------------------ device state
    def _make_cfg(self, embed_dim):
        # `1 - self.memo_momentum` is a Python double before it meets the fp32 embedding (reference :63-65)
        return _lib.VknTrackerCfg(float(self.init_score_thr), float(self.obj_score_thr), float(self.match_score_thr),
                                  float(self.memo_momentum), float(1.0 - self.memo_momentum), float(self.nms_conf_thr),
                                  float(self.nms_backdrop_iou_thr), float(self.nms_class_iou_thr), int(self.memo_tracklet_frames),
                                  int(self.memo_backdrop_frames), int(bool(self.with_cats)), _METRICS[self.match_metric],
                                  self.max_dets, self.max_tracklets, int(embed_dim))

    def _ensure(self, device, embed_dim):
        if self._state is not None and self._device == device and self._cfg.embed_dim == embed_dim:
            return
        if self._state is not None:
            raise ValueError('tracker state exists for another device / embedding size: call reset() first')
        L = _lib.lib()
        cfg = self._make_cfg(embed_dim)
        nb, nw = L.vkn_qd_tracker_state_bytes(ctypes.byref(cfg)), L.vkn_qd_tracker_workspace_bytes(ctypes.byref(cfg))
        if nb == 0:
            raise _lib.VknError(-2, 'tracker capacities outside the supported envelope (max_dets <= 256, '
                                    'max_tracklets + max(memo_backdrop_frames, 1) * max_dets <= 4096, embed_dim <= 1024)')
        self._cfg, self._device = cfg, device
        self._state = torch.empty(nb, dtype=torch.uint8, device=device)
        self._ws = torch.empty(nw, dtype=torch.uint8, device=device)
        off = (ctypes.c_size_t * 12)()
        _lib.check(L.vkn_qd_tracker_state_layout(ctypes.byref(cfg), off))
        self._layout = [int(o) for o in off]
        with torch.cuda.device(device):
            _lib.check(L.vkn_qd_tracker_reset(ctypes.byref(cfg), ops._ptr(self._state), nb, ops._stream()))

    def reset(self):
        """Forget every track (a new video).  The buffers are released; the next `match` re-creates them."""
        self._cfg = self._state = self._ws = self._layout = self._device = None

    # ------------------------------------------------------------------ the per-frame call
    def match_padded(self, bboxes, labels, track_feats, frame_id):
        """-> (out_bboxes [max_dets,5], out_labels [max_dets], out_ids [max_dets] int64, count int32 [2] = (survivors, status)),
        all on the device, nothing synchronised: rows [0, count[0]) are the surviving detections in score order."""
        if not (torch.is_tensor(bboxes) and bboxes.is_cuda and labels.is_cuda and track_feats.is_cuda):
            raise _lib.VknLibraryError('QuasiDenseEmbedTracker: expected CUDA/HIP tensors — the MI355X path has no CPU fallback')
        n = int(bboxes.shape[0])
        if bboxes.dim() != 2 or bboxes.shape[1] != 5 or labels.shape[0] != n or track_feats.shape[0] != n:
            raise ValueError('bboxes [n,5] (x1, y1, x2, y2, score), labels [n], track_feats [n,E]')
        if n > self.max_dets:
            raise ValueError(f'{n} detections > max_dets = {self.max_dets}')
        dev = bboxes.device
        self._ensure(dev, int(track_feats.shape[1]))
        bb = bboxes.detach().to(torch.float32).contiguous()
        lb = labels.detach().to(torch.int64).contiguous()
        em = track_feats.detach().to(torch.float32).contiguous()
        D = self.max_dets
        out_b = torch.empty((D, 5), dtype=torch.float32, device=dev)
        out_l = torch.empty((D,), dtype=torch.int64, device=dev)
        tail = torch.empty((D + 1,), dtype=torch.int64, device=dev)      # ids [D] | (count, status) as two int32
        L = _lib.lib()
        with torch.cuda.device(dev):
            _lib.check(L.vkn_qd_tracker_match_f32(ctypes.byref(self._cfg), ops._ptr(self._state), self._state.numel(), ops._ptr(bb),
                                                  ops._ptr(lb), ops._ptr(em), n, int(frame_id), ops._ptr(out_b), ops._ptr(out_l),
                                                  ops._ptr(tail), ctypes.c_void_p(tail.data_ptr() + 8 * D), ops._ptr(self._ws),
                                                  self._ws.numel(), ops._stream()))
        self._tail = tail
        return out_b, out_l, tail[:D], tail[D:].view(torch.int32)

    def match(self, bboxes, labels, track_feats, frame_id, asso_tau=-1):
        """The reference's signature and returns (:137-207): surviving boxes / labels on the device, `ids` on the host."""
        out_b, out_l, _, _ = self.match_padded(bboxes, labels, track_feats, frame_id)
        host = self._tail.cpu()                                           # ONE small copy: ids + count + status
        k, status = (int(v) for v in host[self.max_dets:].view(torch.int32)[:2])
        if status & 1:
            # reported for THIS frame only (the device state has already advanced: the dropped birth's id is consumed); later
            # frames run normally.  The reference has no cap — raise max_tracklets if this fires.
            raise RuntimeError(f'QuasiDenseEmbedTracker: more than max_tracklets = {self.max_tracklets} live tracks in frame '
                               f'{int(frame_id)}: a birth was dropped')
        return out_b[:k], out_l[:k].to(labels.dtype), host[:k].clone()

    # ------------------------------------------------------------------ the reference's memo surface (host-side views)
    @property
    def memo(self):
        """The reference's `memo` property (:105-135) as a HOST snapshot: dict(bboxes, labels, embeds, ids, vs) over the tracklets in
        creation order followed by the backdrops (ids -1) — what `match` scores detections against.  Introspection only: the match
        kernel reads the device-resident table, never this copy."""
        tr, bd = self.tracklets, self.backdrops
        E = self._cfg.embed_dim if self._cfg is not None else 0
        boxes = [t['bbox'][None] for t in tr.values()] + [b['bboxes'] for b in bd]
        embs = [t['embed'][None] for t in tr.values()] + [b['embeds'] for b in bd]
        ref = next((t['embed'] for t in tr.values()), None)
        if ref is None:
            ref = next((b['embeds'] for b in bd), torch.zeros(0))
        dev, fdt = ref.device, (ref.dtype if ref.is_floating_point() else torch.float32)      # everything on the entries' device / dtype
        labs = [torch.tensor([t['label']], device=dev) for t in tr.values()] + [b['labels'].long().to(dev) for b in bd]
        nb = sum(int(b['bboxes'].shape[0]) for b in bd)
        ids = torch.tensor(list(tr.keys()) + [-1] * nb, dtype=torch.long, device=dev)
        vs = [t['velocity'][None] for t in tr.values()] + [torch.zeros(nb, 5, device=dev, dtype=fdt)]
        cat = lambda xs, shape: torch.cat([x.to(dev) for x in xs]) if xs else torch.zeros(shape, device=dev, dtype=fdt)
        return dict(bboxes=cat(boxes, (0, 5)), labels=cat(labs, (0,)).long(), embeds=cat(embs, (0, E)), ids=ids,
                    vs=cat(vs, (0, 5)))

    def update_memo(self, ids, bboxes, embeds, labels, frame_id):
        """The reference's host-side memo update (:47-103) is part of `vkn_qd_tracker_match_f32` here (births, momentum embeddings,
        velocities, backdrops and expiry happen inside the match kernel, on the device table).  Calling it separately would apply
        the update twice."""
        raise NotImplementedError('update_memo is fused into match(): the memo lives in the device state buffer')

    # ------------------------------------------------------------------ introspection (host copies; not on the hot path)
    def _view(self, idx, dtype, shape):
        n = 1
        for s in shape:
            n *= s
        esz = 4
        return self._state[self._layout[idx]:self._layout[idx] + n * esz].view(dtype).reshape(shape)

    def _header(self):
        if self._state is None:
            return [0] * 16
        return self._view(0, torch.int32, (16,)).cpu().tolist()

    @property
    def num_tracklets(self):
        """Ids handed out so far (the reference's counter, :203)."""
        return self._header()[0]

    @property
    def empty(self):
        return self._header()[1] == 0

    @property
    def status(self):
        return self._header()[3]

    @property
    def tracklets(self):
        """{id: dict(bbox, embed, label, last_frame, velocity, acc_frame)} in creation order — a host snapshot of the device table."""
        if self._state is None:
            return {}
        k = self._header()[1]
        T, E = self.max_tracklets, self._cfg.embed_dim
        ids = self._view(1, torch.int32, (T,))[:k].cpu()
        lab, last, acc = (self._view(i, torch.int32, (T,))[:k].cpu() for i in (2, 3, 4))
        box, vel = (self._view(i, torch.float32, (T, 5))[:k].cpu() for i in (5, 6))
        emb = self._view(7, torch.float32, (T, E))[:k].cpu()
        return {int(ids[i]): dict(bbox=box[i], embed=emb[i], label=int(lab[i]), last_frame=int(last[i]), velocity=vel[i],
                                  acc_frame=int(acc[i])) for i in range(k)}

    @property
    def backdrops(self):
        """[dict(bboxes, embeds, labels)] newest frame first — host snapshot."""
        if self._state is None:
            return []
        F = max(self.memo_backdrop_frames, 1)
        D, E = self.max_dets, self._cfg.embed_dim
        nb = self._header()[2]
        cnt = self._view(8, torch.int32, (F,)).cpu().tolist()
        lab = self._view(9, torch.int32, (F, D)).cpu()
        box = self._view(10, torch.float32, (F, D, 5)).cpu()
        emb = self._view(11, torch.float32, (F, D, E)).cpu()
        return [dict(bboxes=box[f, :cnt[f]], embeds=emb[f, :cnt[f]], labels=lab[f, :cnt[f]]) for f in range(nb)]

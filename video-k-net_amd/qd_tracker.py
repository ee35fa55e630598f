"""`QuasiDenseEmbedTracker` — drop-in for knet/video/qdtrack/trackers/quasi_dense_embed_tracker.py:9-207: the
association step of the video models (`tracker=dict(type='QuasiDenseEmbedTracker', ...)`), fed by the thing boxes
(`vkn_panoptic_joint_f32`'s bbox output) and the tracking embeddings the head already produces.

Same ctor kwargs, `match(bboxes, labels, track_feats, frame_id) -> (bboxes, labels, ids)`, `update_memo`, `memo`, `empty`.  The
work is a [n x m] similarity (n, m <= ~100) and an ORDER-DEPENDENT greedy loop, i.e. host logic in the reference too: the inputs
are brought to the host once per frame (a few KB) and every decision is taken there in the reference's evaluation order, so ids
are bit-identical (tests/golden/qd_tracker.npz).  `bbox_overlaps` restates mmdet 2.18's IoU (third-party).
"""
import torch
import torch.nn.functional as F

from .registry import Registry

TRACKERS = Registry('tracker')


def build_tracker(cfg):
    return TRACKERS.build(cfg)


def bbox_overlaps(b1, b2, eps=1e-6):
    """mmdet.core.bbox_overlaps(mode='iou', is_aligned=False) on [x1, y1, x2, y2] boxes."""
    rows, cols = b1.size(0), b2.size(0)
    if rows * cols == 0:
        return b1.new_zeros((rows, cols))
    a1 = (b1[:, 2] - b1[:, 0]) * (b1[:, 3] - b1[:, 1])
    a2 = (b2[:, 2] - b2[:, 0]) * (b2[:, 3] - b2[:, 1])
    lt = torch.max(b1[:, None, :2], b2[None, :, :2])
    rb = torch.min(b1[:, None, 2:], b2[None, :, 2:])
    wh = (rb - lt).clamp(min=0)
    overlap = wh[..., 0] * wh[..., 1]
    union = torch.max(a1[:, None] + a2[None, :] - overlap, overlap.new_tensor([eps]))
    return overlap / union


@TRACKERS.register_module()
class QuasiDenseEmbedTracker:

    def __init__(self, init_score_thr=0.8, obj_score_thr=0.5, match_score_thr=0.5, memo_tracklet_frames=10, memo_backdrop_frames=1,
                 memo_momentum=0.8, nms_conf_thr=0.5, nms_backdrop_iou_thr=0.3, nms_class_iou_thr=0.7, with_cats=True,
                 match_metric='bisoftmax'):
        assert 0 <= memo_momentum <= 1.0 and memo_tracklet_frames >= 0 and memo_backdrop_frames >= 0
        assert match_metric in ['bisoftmax', 'softmax', 'cosine']
        self.init_score_thr, self.obj_score_thr, self.match_score_thr = init_score_thr, obj_score_thr, match_score_thr
        self.memo_tracklet_frames, self.memo_backdrop_frames, self.memo_momentum = memo_tracklet_frames, memo_backdrop_frames, memo_momentum
        self.nms_conf_thr, self.nms_backdrop_iou_thr, self.nms_class_iou_thr = nms_conf_thr, nms_backdrop_iou_thr, nms_class_iou_thr
        self.with_cats, self.match_metric = with_cats, match_metric
        self.num_tracklets = 0
        self.tracklets = dict()
        self.backdrops = []

    @property
    def empty(self):
        return False if self.tracklets else True

    def update_memo(self, ids, bboxes, embeds, labels, frame_id):                                      # reference :47-103
        tracklet_inds = ids > -1
        for id_, bbox, embed, label in zip(ids[tracklet_inds], bboxes[tracklet_inds], embeds[tracklet_inds], labels[tracklet_inds]):
            id_ = int(id_)
            if id_ in self.tracklets:
                t = self.tracklets[id_]
                velocity = (bbox - t['bbox']) / (frame_id - t['last_frame'])
                t['bbox'] = bbox
                t['embed'] = (1 - self.memo_momentum) * t['embed'] + self.memo_momentum * embed
                t['last_frame'] = frame_id
                t['label'] = label
                t['velocity'] = (t['velocity'] * t['acc_frame'] + velocity) / (t['acc_frame'] + 1)
                t['acc_frame'] += 1
            else:
                self.tracklets[id_] = dict(bbox=bbox, embed=embed, label=label, last_frame=frame_id,
                                           velocity=torch.zeros_like(bbox), acc_frame=0)
        backdrop_inds = torch.nonzero(ids == -1, as_tuple=False).squeeze(1)
        ious = bbox_overlaps(bboxes[backdrop_inds, :-1], bboxes[:, :-1])
        for i, ind in enumerate(backdrop_inds):
            if (ious[i, :ind] > self.nms_backdrop_iou_thr).any():
                backdrop_inds[i] = -1
        backdrop_inds = backdrop_inds[backdrop_inds > -1]
        self.backdrops.insert(0, dict(bboxes=bboxes[backdrop_inds], embeds=embeds[backdrop_inds], labels=labels[backdrop_inds]))
        for k in [k for k, v in self.tracklets.items() if frame_id - v['last_frame'] >= self.memo_tracklet_frames]:
            self.tracklets.pop(k)
        if len(self.backdrops) > self.memo_backdrop_frames:
            self.backdrops.pop()

    @property
    def memo(self):                                                                                    # reference :105-135
        memo_embeds, memo_ids, memo_bboxes, memo_labels, memo_vs = [], [], [], [], []
        for k, v in self.tracklets.items():
            memo_bboxes.append(v['bbox'][None, :])
            memo_embeds.append(v['embed'][None, :])
            memo_ids.append(k)
            memo_labels.append(v['label'].view(1, 1))
            memo_vs.append(v['velocity'][None, :])
        memo_ids = torch.tensor(memo_ids, dtype=torch.long).view(1, -1)
        for backdrop in self.backdrops:
            memo_bboxes.append(backdrop['bboxes'])
            memo_embeds.append(backdrop['embeds'])
            memo_ids = torch.cat([memo_ids, torch.full((1, backdrop['embeds'].size(0)), -1, dtype=torch.long)], dim=1)
            memo_labels.append(backdrop['labels'][:, None])
            memo_vs.append(torch.zeros_like(backdrop['bboxes']))
        return (torch.cat(memo_bboxes, dim=0), torch.cat(memo_labels, dim=0).squeeze(1), torch.cat(memo_embeds, dim=0),
                memo_ids.squeeze(0), torch.cat(memo_vs, dim=0))

    def match(self, bboxes, labels, track_feats, frame_id, asso_tau=-1):                              # reference :137-207
        dev = bboxes.device
        bboxes, labels, track_feats = bboxes.detach().cpu(), labels.detach().cpu(), track_feats.detach().cpu()   # one small D2H
        _, inds = bboxes[:, -1].sort(descending=True)
        bboxes, labels, embeds = bboxes[inds, :], labels[inds], track_feats[inds, :]
        valids = bboxes.new_ones((bboxes.size(0)))
        ious = bbox_overlaps(bboxes[:, :-1], bboxes[:, :-1])
        for i in range(1, bboxes.size(0)):
            thr = self.nms_backdrop_iou_thr if bboxes[i, -1] < self.obj_score_thr else self.nms_class_iou_thr
            if (ious[i, :i] > thr).any():
                valids[i] = 0
        valids = valids == 1
        bboxes, labels, embeds = bboxes[valids, :], labels[valids], embeds[valids, :]
        ids = torch.full((bboxes.size(0),), -1, dtype=torch.long)
        if bboxes.size(0) > 0 and not self.empty:
            memo_bboxes, memo_labels, memo_embeds, memo_ids, memo_vs = self.memo
            if self.match_metric == 'bisoftmax':
                feats = torch.mm(embeds, memo_embeds.t())
                scores = (feats.softmax(dim=1) + feats.softmax(dim=0)) / 2
            elif self.match_metric == 'softmax':
                scores = torch.mm(embeds, memo_embeds.t()).softmax(dim=1)
            else:
                scores = torch.mm(F.normalize(embeds, p=2, dim=1), F.normalize(memo_embeds, p=2, dim=1).t())
            if self.with_cats:
                scores *= (labels.view(-1, 1) == memo_labels.view(1, -1)).float()
            for i in range(bboxes.size(0)):
                conf, memo_ind = torch.max(scores[i, :], dim=0)
                id_ = memo_ids[memo_ind]
                if conf > self.match_score_thr:
                    if id_ > -1:
                        if bboxes[i, -1] > self.obj_score_thr:
                            ids[i] = id_
                            scores[:i, memo_ind] = 0
                            scores[i + 1:, memo_ind] = 0
                        elif conf > self.nms_conf_thr:
                            ids[i] = -2
        new_inds = (ids == -1) & (bboxes[:, 4] > self.init_score_thr)
        num_news = int(new_inds.sum())
        ids[new_inds] = torch.arange(self.num_tracklets, self.num_tracklets + num_news, dtype=torch.long)
        self.num_tracklets += num_news
        self.update_memo(ids, bboxes, embeds, labels, frame_id)
        return bboxes.to(dev), labels.to(dev), ids

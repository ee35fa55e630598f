"""ORACLE — TEST INFRASTRUCTURE ONLY (never imported by the product package).

CPU restatement of the quasi-dense embedding association of the video models — what the reference's
`QuasiDenseEmbedTracker.match` decides per frame (knet/video/qdtrack/trackers/quasi_dense_embed_tracker.py:137-207, with the
memo bookkeeping of `update_memo` :47-103 and `memo` :105-135) — written as a flat slot table (the layout of the device
kernel it checks, csrc/vkn_tracker.hip), float32 tensors, torch CPU ops for the matrix product / softmax so that the values
that feed decisions come from the same ATen kernels the reference uses.

Pinned by tests/golden/qd_tracker.npz: ids / labels / boxes of the reference's own tracker class on four synthetic videos
(oracle/gen_golden_tracker.py), bit for bit (tests/test_oracle_golden.py).
"""
import numpy as np
import torch
import torch.nn.functional as F


def iou_matrix(a, b, eps=1e-6):
    """mmdet 2.18 `bbox_overlaps(a, b, mode='iou')` on [x1, y1, x2, y2] rows (third-party; restated)."""
    if a.shape[0] * b.shape[0] == 0:
        return a.new_zeros((a.shape[0], b.shape[0]))
    area_a = (a[:, 2] - a[:, 0]) * (a[:, 3] - a[:, 1])
    area_b = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    wh = (torch.min(a[:, None, 2:], b[None, :, 2:]) - torch.max(a[:, None, :2], b[None, :, :2])).clamp(min=0)
    inter = wh[..., 0] * wh[..., 1]
    return inter / torch.max(area_a[:, None] + area_b[None, :] - inter, inter.new_tensor([eps]))


class TrackerOracle:
    """State: parallel lists, one entry per live track, in creation order; `backdrops`: list of frames, newest first."""

    def __init__(self, init_score_thr=0.8, obj_score_thr=0.5, match_score_thr=0.5, memo_tracklet_frames=10, memo_backdrop_frames=1,
                 memo_momentum=0.8, nms_conf_thr=0.5, nms_backdrop_iou_thr=0.3, nms_class_iou_thr=0.7, with_cats=True,
                 match_metric='bisoftmax'):
        self.p = dict(init=init_score_thr, obj=obj_score_thr, match=match_score_thr, keep_frames=memo_tracklet_frames,
                      bd_frames=memo_backdrop_frames, mom=memo_momentum, conf=nms_conf_thr, bd_iou=nms_backdrop_iou_thr,
                      cls_iou=nms_class_iou_thr, cats=with_cats, metric=match_metric)
        self.next_id = 0
        self.t_id, self.t_label, self.t_last, self.t_acc = [], [], [], []
        self.t_box, self.t_vel, self.t_emb = [], [], []
        self.backdrops = []

    def step(self, boxes, labels, embeds, frame_id):
        """boxes [n,5] fp32, labels [n] int64, embeds [n,E] fp32 (torch CPU) -> (boxes [k,5], labels [k], ids [k] int64)."""
        p = self.p
        order = torch.sort(boxes[:, 4], descending=True)[1]                                   # :139-142
        boxes, labels, embeds = boxes[order], labels[order], embeds[order]
        n = boxes.shape[0]
        ov = iou_matrix(boxes[:, :4], boxes[:, :4])
        alive = torch.ones(n, dtype=torch.bool)
        for i in range(1, n):                                                                  # :146-152: against EVERY better box
            limit = p['bd_iou'] if boxes[i, 4] < p['obj'] else p['cls_iou']
            alive[i] = not bool((ov[i, :i] > limit).any())
        boxes, labels, embeds = boxes[alive], labels[alive], embeds[alive]
        k = boxes.shape[0]
        ids = torch.full((k,), -1, dtype=torch.long)
        slot = [-1] * k
        if k > 0 and self.t_id:                                                                # :162 (`not self.empty`)
            m_emb = torch.stack(self.t_emb) if self.t_emb else embeds.new_zeros((0, embeds.shape[1]))
            m_id = list(self.t_id)
            m_label = [int(v) for v in self.t_label]
            for fr in self.backdrops:                                                          # :121-129
                m_emb = torch.cat([m_emb, fr['emb']])
                m_id += [-1] * fr['emb'].shape[0]
                m_label += [int(v) for v in fr['label']]
            if p['metric'] == 'cosine':                                                        # :173-176
                sim = torch.mm(F.normalize(embeds, p=2, dim=1), F.normalize(m_emb, p=2, dim=1).t())
            else:
                raw = torch.mm(embeds, m_emb.t())
                sim = raw.softmax(dim=1)
                if p['metric'] == 'bisoftmax':                                                 # :164-168
                    sim = (sim + raw.softmax(dim=0)) / 2
            if p['cats']:                                                                      # :178-180
                sim = sim * (labels.view(-1, 1) == torch.tensor(m_label).view(1, -1)).float()
            taken = torch.zeros(sim.shape[1], dtype=torch.bool)
            for i in range(k):                                                                 # :182-196
                row = torch.where(taken, torch.zeros(()), sim[i])
                conf, j = torch.max(row, dim=0)
                j = int(j)
                if conf > p['match'] and m_id[j] > -1:
                    if boxes[i, 4] > p['obj']:
                        ids[i], slot[i] = m_id[j], j
                        taken[j] = True
                    elif conf > p['conf']:
                        ids[i] = -2
        born = (ids == -1) & (boxes[:, 4] > p['init'])                                         # :197-203
        ids[born] = torch.arange(self.next_id, self.next_id + int(born.sum()), dtype=torch.long)
        self.next_id += int(born.sum())
        for i in range(k):                                                                     # :50-79
            if slot[i] >= 0:
                t = slot[i]
                vel = (boxes[i] - self.t_box[t]) / (frame_id - self.t_last[t])
                self.t_vel[t] = (self.t_vel[t] * self.t_acc[t] + vel) / (self.t_acc[t] + 1)
                self.t_box[t] = boxes[i]
                self.t_emb[t] = (1 - p['mom']) * self.t_emb[t] + p['mom'] * embeds[i]
                self.t_label[t], self.t_last[t] = labels[i], frame_id
                self.t_acc[t] += 1
            elif bool(born[i]):
                self.t_id.append(int(ids[i])); self.t_label.append(labels[i]); self.t_last.append(frame_id); self.t_acc.append(0)
                self.t_box.append(boxes[i]); self.t_vel.append(torch.zeros_like(boxes[i])); self.t_emb.append(embeds[i])
        cand = [i for i in range(k) if int(ids[i]) == -1]                                      # :81-93
        ov = iou_matrix(boxes[:, :4], boxes[:, :4])
        cand = [i for i in cand if not bool((ov[i, :i] > p['bd_iou']).any())]
        sel = torch.tensor(cand, dtype=torch.long)
        self.backdrops.insert(0, dict(box=boxes[sel], emb=embeds[sel], label=labels[sel]))
        keep = [t for t in range(len(self.t_id)) if frame_id - self.t_last[t] < p['keep_frames']]   # :95-100
        for name in ('t_id', 't_label', 't_last', 't_acc', 't_box', 't_vel', 't_emb'):
            setattr(self, name, [getattr(self, name)[t] for t in keep])
        del self.backdrops[p['bd_frames']:]                                                    # :102-103
        return boxes, labels, ids


def random_video(T, n_max, emb, n_cls, seed):
    """Dense random detections for property tests beyond the golden sizes: n_max objects with persistent codes, jittered boxes,
    random visibility and scores (numpy RNG: inputs only, no reference output depends on them)."""
    rng = np.random.default_rng(seed)
    codes = rng.standard_normal((n_max, emb)).astype(np.float32) * 2.0
    cls = rng.integers(0, n_cls, n_max)
    ctr = rng.uniform(50, 1500, (n_max, 2))
    size = rng.uniform(15, 80, (n_max, 2))
    vel = rng.uniform(-8, 8, (n_max, 2))
    frames = []
    for t in range(T):
        vis = np.nonzero(rng.uniform(size=n_max) < 0.8)[0]
        rng.shuffle(vis)
        c = ctr[vis] + vel[vis] * t + rng.normal(0, 1.0, (len(vis), 2))
        sc = rng.uniform(0.05, 0.99, len(vis))
        boxes = np.concatenate([c - size[vis], c + size[vis], sc[:, None]], 1).astype(np.float32)
        embs = (codes[vis] + rng.standard_normal((len(vis), emb)).astype(np.float32) * 0.4).astype(np.float32)
        frames.append((boxes, cls[vis].astype(np.int64), embs))
    return frames

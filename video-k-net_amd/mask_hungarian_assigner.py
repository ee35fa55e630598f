"""`MaskHungarianAssigner` — drop-in for the reference's train-time assigner (knet/det/mask_hungarian_assigner.py:117-274):
same ctor kwargs and `assign(...)` signature / result fields.  The [N x G] cost matrix is computed on the GPU
(`vkn_assign_costs_f32`: the mask contractions run on the gather kernel), the linear sum assignment in libvkn's C++
shortest-augmenting-path solver (`vkn_lsap_f32`, the algorithm scipy uses) — scipy is not needed at run time.

Supported: the costs every shipped config uses — `FocalLossCost` (mmdet 2.18 defaults), `DiceCost(pred_act=True)`,
`MaskCost(pred_act=True)`, `topk=1`, no boundary cost.  Anything else raises NotImplementedError.
"""
import weakref

import numpy as np
import torch

from . import ops


class AssignResult:
    """Fields of mmdet.core.AssignResult that the reference reads (knet/det/mask_pseudo_sampler.py)."""

    def __init__(self, num_gts, gt_inds, max_overlaps, labels=None):
        self.num_gts = num_gts
        self.gt_inds = gt_inds
        self.max_overlaps = max_overlaps
        self.labels = labels
        self.host_pos_inds = None   # sorted numpy array of the matched prediction indices when the assigner knows them on the host
        self.device_pos_inds = None # the same as a device tensor (device LSAP): its length is min(N, G), no synchronisation needed
        self.status = None


class DeviceAssignResult(AssignResult):
    """The result of a device assignment of `assign_batch`: `gt_inds` (what the reference's loop and its tests read) exists; `labels`
    (per-prediction label, -1 = none) and `device_pos_inds` (int64) are three small launches per image that only the sampler path
    reads — built on first access.  `_pairs32` = the LSAP's own (row, col) int32 pairs, what the fused loss tail consumes."""

    def __init__(self, num_gts, gt_inds, pairs32, gt_labels, status):
        self.num_gts, self.gt_inds, self.max_overlaps = num_gts, gt_inds, None
        self._pairs32, self._gt_labels = pairs32, gt_labels
        self._labels = self._pos = None
        self.host_pos_inds, self.status = None, status

    @property
    def device_pos_inds(self):
        if self._pos is None:
            self._pos = self._pairs32[0].long()
        return self._pos

    @device_pos_inds.setter
    def device_pos_inds(self, value):
        self._pos = value

    @property
    def labels(self):
        if self._labels is None:
            n = self.gt_inds.shape[0]
            lab = self.gt_inds.new_full((n,), -1, dtype=torch.long)
            lab[self.device_pos_inds] = self._gt_labels.to(device=lab.device, dtype=lab.dtype)[self._pairs32[1].long()]
            self._labels = lab
        return self._labels

    @labels.setter
    def labels(self, value):       # (mmdet's AssignResult is a plain attribute bag: callers may overwrite)
        self._labels = value


class _AsyncFlags:
    """Error flags computed on the device and read WITHOUT stalling the stream: `push` enqueues a copy of the flag into pinned host
    memory and records an event; `poll` looks only at flags whose event has completed (`wait=True`: all of them).  The training loop
    polls at the start of the next step — an invalid cost matrix or label is reported one step late instead of costing a
    synchronisation per step."""

    def __init__(self):
        self.items = []

    def push(self, flag, message):
        host = torch.empty((), dtype=torch.int32).pin_memory()
        host.copy_(flag.to(torch.int32).reshape(()), non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self.items.append((ev, host, message))
        if len(self.items) > 256:            # never drop an unread flag silently: read the oldest ones now (their events are long done)
            old, self.items = self.items[:-256], self.items[-256:]
            for ev_o, host_o, msg_o in old:
                ev_o.synchronize()
                if int(host_o) != 0:
                    raise (IndexError if 'gt_labels' in msg_o else ValueError)(msg_o)

    def poll(self, wait=False):
        keep, bad = [], None
        for ev, host, message in self.items:
            if wait:
                ev.synchronize()
            if ev.query():
                if int(host) != 0 and bad is None:
                    bad = message
            else:
                keep.append((ev, host, message))
        self.items = keep
        if bad is not None:
            raise (IndexError if 'gt_labels' in bad else ValueError)(bad)


FLAGS = _AsyncFlags()     # one queue per process: the assigners of all stages share it


def _cost_cfg(cfg, kind, allowed, defaults):
    cfg = dict(cfg or {})
    typ = cfg.pop('type', None)
    if typ not in allowed:
        raise NotImplementedError(f'{kind}: type {typ!r} is not provided (shipped configs use {allowed})')
    out = dict(defaults)
    for k, v in cfg.items():
        if k not in out:
            raise NotImplementedError(f'{kind}: option {k!r} is not provided')
        out[k] = v
    return out


class MaskHungarianAssigner:
    # lower clamp of sigmoid(mask logits) inside (DiceCost, MaskCost): knet/det/mask_hungarian_assigner.py:69,101.  The knet_vis copies
    # of the two cost classes (same registry names) do not clamp — `pred_clamp = (0.0, 0.0)` selects that flavour
    pred_clamp = (1e-3, 1e-2)
    lowres_costs = True    # `assign_batch(.., lowres=..)`: costs straight from the low-res logits (False: from the up-scaled tensors, A/B)

    def __init__(self, cls_cost=dict(type='ClassificationCost', weight=1.), mask_cost=dict(type='SigmoidCost', weight=1.0),
                 dice_cost=dict(), boundary_cost=None, topk=1):
        if boundary_cost is not None:
            raise NotImplementedError('boundary_cost is not provided (no shipped config sets it)')
        if topk != 1:
            raise NotImplementedError('topk > 1 is not provided (no shipped config sets it)')
        self.cls = _cost_cfg(cls_cost, 'cls_cost', ('FocalLossCost',), dict(weight=1.0, alpha=0.25, gamma=2, eps=1e-12))
        self.mask = _cost_cfg(mask_cost, 'mask_cost', ('MaskCost',), dict(weight=1.0, pred_act=False, act_mode='sigmoid'))
        self.dice = _cost_cfg(dice_cost, 'dice_cost', ('DiceCost',), dict(weight=1.0, pred_act=False, act_mode='sigmoid', eps=1e-3))
        for c, name in ((self.mask, 'MaskCost'), (self.dice, 'DiceCost')):
            if not c['pred_act'] or c['act_mode'] != 'sigmoid':
                raise NotImplementedError(f'{name} needs pred_act=True, act_mode="sigmoid" (every shipped config)')
        self.topk = topk
        self.lsap = 'device'        # 'host': copy the cost matrix to the host and run vkn_lsap_f32 there (the round-1/2 path)
        self.pending_status = []

    @staticmethod
    def _label_key(gt_labels, ncls):
        return (gt_labels.data_ptr(), gt_labels._version, tuple(gt_labels.shape), str(gt_labels.device), int(ncls))

    @classmethod
    def validate_labels(cls, label_tensors, ncls, status=None):
        """Range-check the ground-truth labels of a whole batch against the `ncls` class logits ONCE per step, on the device: the result
        goes to the asynchronous flag queue (`FLAGS`), `assign` skips its own per-image check.  An out-of-range label is what makes the
        reference's `cls_pred[:, gt_labels]` raise an IndexError; here the cost kernel clamps the index (no stray read) and the error
        surfaces at the next `FLAGS.poll()` — one step later, without a device -> host synchronisation in the step.  CPU label
        tensors are checked on the spot.  `status` (device int32 [1]): accumulate the verdict there (bit 1) instead of queueing a
        flag — the caller hands the word to `FLAGS` itself (the fused training tail: one word per step)."""
        cls._validated.clear()        # (keys name storage: they are only trusted for the step that checked them)
        word = status
        for t in label_tensors:
            if not torch.is_tensor(t) or t.numel() == 0:
                continue
            if t.is_cuda:
                if word is None:
                    word = torch.zeros(1, dtype=torch.int32, device=t.device)
                ops.check_range(t, 0, ncls, 2, word)          # one small launch per tensor: *word |= 2 when a label is out of range
            else:
                lo, hi = int(t.min()), int(t.max())
                if lo < 0 or hi >= ncls:
                    raise IndexError(f'gt_labels outside [0, {ncls}): {lo} .. {hi}')
            cls._validated[cls._label_key(t, ncls)] = weakref.ref(t)    # identity, not address: a recycled allocation must not match
        if word is not None and status is None:
            FLAGS.push(word, f'gt_labels outside [0, {ncls})')

    _validated = {}

    @classmethod
    def _is_validated(cls, t, ncls):
        r = cls._validated.get(cls._label_key(t, ncls)) if torch.is_tensor(t) else None
        return r is not None and r() is t

    def cost_matrix(self, bbox_pred, cls_pred, gt_bboxes, gt_labels):
        """[N, G] device tensor: cls_cost + mask_cost + dice_cost (reference :222-241)."""
        use_cls = self.cls['weight'] != 0 and cls_pred is not None
        checked = use_cls and self._is_validated(gt_labels, cls_pred.shape[1])
        return ops.assign_costs(bbox_pred, cls_pred if use_cls else None, gt_bboxes, gt_labels, labels_checked=checked,
                                cls_weight=self.cls['weight'] if use_cls else 0.0, dice_weight=self.dice['weight'],
                                mask_weight=self.mask['weight'], focal_alpha=self.cls['alpha'], focal_gamma=self.cls['gamma'],
                                focal_eps=self.cls['eps'], dice_eps=self.dice['eps'], dice_pred_min=self.pred_clamp[0],
                                mask_pred_min=self.pred_clamp[1])

    def assign(self, bbox_pred, cls_pred, gt_bboxes, gt_labels, img_meta=None, gt_bboxes_ignore=None, eps=1e-7):
        """bbox_pred = mask logits [N,H,W], gt_bboxes = gt masks [G,H,W] (the reference's argument names).
        -> AssignResult(num_gts, gt_inds [N] (0 = background, else 1-based gt index), None, labels [N] (-1 = none))."""
        assert gt_bboxes_ignore is None, 'Only case when gt_bboxes_ignore is None is supported.'
        num_gts, num_bboxes = gt_bboxes.size(0), bbox_pred.size(0)
        gt_inds = bbox_pred.new_full((num_bboxes,), -1, dtype=torch.long)
        labels = bbox_pred.new_full((num_bboxes,), -1, dtype=torch.long)
        if num_gts == 0 or num_bboxes == 0:
            if num_gts == 0:
                gt_inds[:] = 0
            return AssignResult(num_gts, gt_inds, None, labels=labels)
        cost = self.cost_matrix(bbox_pred, cls_pred, gt_bboxes, gt_labels)
        if self.lsap == 'device' and num_gts <= 256 and num_bboxes <= 256:
            # the whole assignment stays on the device and on the stream: no copy, no synchronisation (vkn_lsap_batch_f32)
            gts, rows, cols, status = ops.lsap_device([cost])
            rows, cols = rows[0].long(), cols[0].long()
            gt_inds = gts[0]
            labels[rows] = gt_labels.to(device=labels.device, dtype=labels.dtype)[cols]
            res = AssignResult(num_gts, gt_inds, None, labels=labels)
            res.device_pos_inds = rows                        # sorted; their number min(N, G) is known without asking the device
            res.status = status
            self._remember(status)
            return res
        rows, cols = ops.lsap(cost)                       # one D2H copy of [N, G] floats, C++ solver on the host
        rows_host = rows
        rows = torch.from_numpy(rows).to(bbox_pred.device)
        cols = torch.from_numpy(cols).to(bbox_pred.device)
        gt_inds[:] = 0
        gt_inds[rows] = cols + 1
        labels[rows] = gt_labels[cols].to(labels.dtype)
        res = AssignResult(num_gts, gt_inds, None, labels=labels)
        res.host_pos_inds = np.sort(np.asarray(rows_host, dtype=np.int64))   # the LSAP ran on the host: the sampler needs no device nonzero
        return res

    def assign_batch(self, bbox_preds, cls_preds, gt_bboxes, gt_labels, img_metas=None, lowres=None):
        """`assign` for the images of a batch (lists of per-image tensors; `cls_preds` entries may be None): the cost matrices are
        computed image by image, ALL linear sum assignments run in ONE launch (one wavefront per image, vkn_lsap_batch_f32).
        -> list of AssignResult, identical to calling `assign` per image.
        `lowres = (per-image [N, h, w] logits, stride)`: `bbox_preds` ARE their x`stride` bilinear up-scaling (the caller vouches) —
        the costs then come from the low-res logits in one kernel for the batch (vkn_assign_costs_lowres_batch_f32) and the up-scaled
        tensors are not read."""
        n = len(bbox_preds)
        if self.lsap != 'device' or any(g.size(0) == 0 or g.size(0) > 256 or p.size(0) == 0 or p.size(0) > 256
                                        for g, p in zip(gt_bboxes, bbox_preds)):
            return [self.assign(bbox_preds[i], cls_preds[i], gt_bboxes[i], gt_labels[i],
                                img_meta=img_metas[i] if img_metas is not None else None) for i in range(n)]
        use_cls = self.cls['weight'] != 0 and all(c is not None for c in cls_preds)
        same = len({p.shape for p in bbox_preds}) == 1 and (not use_cls or len({c.shape for c in cls_preds}) == 1)
        checked = not use_cls or all(self._is_validated(l, cls_preds[0].shape[1]) for l in gt_labels)
        kw = dict(cls_weight=self.cls['weight'] if use_cls else 0.0, dice_weight=self.dice['weight'], mask_weight=self.mask['weight'],
                  focal_alpha=self.cls['alpha'], focal_gamma=self.cls['gamma'], focal_eps=self.cls['eps'], dice_eps=self.dice['eps'],
                  dice_pred_min=self.pred_clamp[0], mask_pred_min=self.pred_clamp[1])
        batched = same and checked and (use_cls or all(c is None for c in cls_preds) or self.cls['weight'] == 0)
        if lowres is not None and self.lowres_ready(lowres[0], lowres[1], cls_preds, gt_bboxes, gt_labels):
            costs = ops.assign_costs_lowres_batch(lowres[0], lowres[1], cls_preds if use_cls else None, gt_bboxes, gt_labels, **kw)
        elif batched:
            # every image's cost matrix from ONE C call (the labels were range-checked by validate_labels)
            costs = ops.assign_costs_batch(bbox_preds, cls_preds if use_cls else None, gt_bboxes, gt_labels, **kw)
        else:
            costs = [self.cost_matrix(bbox_preds[i], cls_preds[i], gt_bboxes[i], gt_labels[i]) for i in range(n)]
        gts, rows, cols, status = ops.lsap_device(costs)
        self._remember(status)
        return [DeviceAssignResult(gt_bboxes[i].size(0), gts[i], (rows[i], cols[i]), gt_labels[i], status) for i in range(n)]

    def lowres_ready(self, lows, stride, cls_preds, gt_bboxes, gt_labels):
        """Will `assign_batch(.., lowres=(lows, stride))` / `assign_batch_lowres` take the costs from the low-res logits?  (every
        condition of the batched device path + the kernel's shape gate) — a caller that gets True need not up-scale at all."""
        if not self.lowres_costs or self.lsap != 'device' or not lows:
            return False
        if any(g.dim() != 3 or g.size(0) == 0 or g.size(0) > 256 for g in gt_bboxes) or len(gt_bboxes) != len(lows):
            return False
        if any(not (l.is_cuda and l.dtype == torch.float32 and l.dim() == 3) for l in lows) or len({l.shape for l in lows}) != 1:
            return False
        N, h, w = lows[0].shape
        if N == 0 or N > 256 or any(tuple(g.shape[1:]) != (stride * h, stride * w) for g in gt_bboxes):
            return False
        use_cls = self.cls['weight'] != 0 and all(c is not None for c in cls_preds)
        if use_cls and (len({c.shape for c in cls_preds}) != 1 or not all(self._is_validated(l, cls_preds[0].shape[1]) for l in gt_labels)):
            return False
        if not use_cls and self.cls['weight'] != 0 and any(c is not None for c in cls_preds):
            return False
        return ops.assign_costs_lowres_supported(N, [int(g.shape[0]) for g in gt_bboxes], h, w, stride)

    def assign_batch_lowres(self, lows, stride, cls_preds, gt_bboxes, gt_labels):
        """`assign_batch` on the low-res logits alone (`lowres_ready(..)` must hold): the x`stride` up-scaled predictions the reference
        assigns on (knet/det/kernel_iter_head.py:150-156) are never formed."""
        if not self.lowres_ready(lows, stride, cls_preds, gt_bboxes, gt_labels):
            raise ValueError('assign_batch_lowres: lowres_ready() does not hold for these inputs')
        use_cls = self.cls['weight'] != 0 and all(c is not None for c in cls_preds)
        costs = ops.assign_costs_lowres_batch(lows, stride, cls_preds if use_cls else None, gt_bboxes, gt_labels,
                                              cls_weight=self.cls['weight'] if use_cls else 0.0, dice_weight=self.dice['weight'],
                                              mask_weight=self.mask['weight'], focal_alpha=self.cls['alpha'], focal_gamma=self.cls['gamma'],
                                              focal_eps=self.cls['eps'], dice_eps=self.dice['eps'], dice_pred_min=self.pred_clamp[0],
                                              mask_pred_min=self.pred_clamp[1])
        gts, rows, cols, status = ops.lsap_device(costs)
        self._remember(status)
        return [DeviceAssignResult(gt_bboxes[i].size(0), gts[i], (rows[i], cols[i]), gt_labels[i], status) for i in range(len(lows))]

    def _remember(self, status):
        """Queue a device status tensor for the next `check_status`.  The list is bounded by FOLDING the oldest entries into one
        `any()` word, never by dropping them: a head that polls rarely still sees every failure."""
        self.pending_status.append(status)
        if len(self.pending_status) > 64:
            head, self.pending_status = self.pending_status[:-32], self.pending_status[-32:]
            self.pending_status.insert(0, torch.cat([h.reshape(-1) for h in head]).any().to(torch.int32).reshape(1))

    def check_status(self, *others, wait=True):
        """Hand the status words of the device assignments issued since the last call by this assigner (and `others`: the per-stage
        assigners of a head) to the asynchronous flag queue and poll it.  `wait=True` (default): block until every queued flag is
        known — raises like the host solver does for NaN / -inf entries or an infeasible matrix.  `wait=False` (the training loop):
        only flags that are already on the host are looked at; nothing stalls, an error surfaces at a later poll."""
        pend = []
        for a in (self,) + tuple(others):
            if hasattr(a, 'pending_status'):
                pend += a.pending_status
                a.pending_status = []
        if pend:
            FLAGS.push(torch.cat(pend).any(), 'linear sum assignment: the cost matrix holds invalid entries or is infeasible')
        FLAGS.poll(wait=wait)


class MaskHungarianAssignerVideo(MaskHungarianAssigner):
    """Clip-level assigner of the VIS tracker head (knet_vis/tracker/mask_hungarian_assigner.py:17-190): one ground-truth unit is
    an INSTANCE OF THE CLIP — its masks in the frames where it appears, zeros elsewhere — and the costs are those of the per-frame
    assigner on "tall" masks, the F frames of a clip stacked along the row axis ([Q, F*H, W] against [G, F*H, W]).  So this is the
    same GPU cost kernel and the same LSAP on a different layout; what is added is the construction of the clip instances.
    knet_vis's DiceCost / MaskCost take the plain sigmoid (knet_vis/det/mask_hungarian_assigner.py:69,100): no clamp."""
    pred_clamp = (0.0, 0.0)

    @staticmethod
    def clip_instances(num_frames, gt_masks, gt_labels, gt_instance_ids):
        """gt_masks: per frame [n_f, H, W]; gt_labels / gt_instance_ids: [M, 2] rows (frame, label) / (frame, instance id), the rows
        of one frame in the order of that frame's masks.  -> (clip masks [G, F, H, W] float, labels [G] long, instance ids [G]),
        instances in ascending id order (reference :104-128).  The bookkeeping runs on the host on the two tiny id tables; the
        masks move with ONE gather on their own device."""
        ids = np.asarray(gt_instance_ids.detach().cpu().numpy(), dtype=np.int64).reshape(-1, 2)
        lab = np.asarray(gt_labels.detach().cpu().numpy(), dtype=np.int64).reshape(-1, 2)
        dev = gt_masks[0].device
        H, W = gt_masks[0].shape[-2:]
        inst = np.unique(ids[:, 1])
        G = len(inst)
        if G == 0:
            return gt_masks[0].new_zeros((0, num_frames, H, W), dtype=torch.float32), torch.zeros(0, dtype=torch.long, device=dev), inst
        base = np.concatenate([[0], np.cumsum([int(m.shape[0]) for m in gt_masks])])       # row of a frame's first mask in the stack
        src, dst = [], []
        labels = np.full(G, -1, dtype=np.int64)
        for f in range(num_frames):
            rows = np.nonzero(ids[:, 0] == f)[0]
            lrows = np.nonzero(lab[:, 0] == f)[0]
            if len(rows) != int(gt_masks[f].shape[0]) or len(lrows) != len(rows):
                raise ValueError(f'frame {f}: {gt_masks[f].shape[0]} masks, {len(rows)} instance ids, {len(lrows)} labels')
            g = np.searchsorted(inst, ids[rows, 1])
            if len(np.unique(g)) != len(g):
                raise ValueError(f'frame {f}: an instance id appears twice')
            src.append(base[f] + np.arange(len(rows)))
            dst.append(g * num_frames + f)
            fl = lab[lrows, 1]
            clash = (labels[g] >= 0) & (labels[g] != fl)
            if clash.any():
                raise ValueError(f'instance {inst[g[clash][0]]} changes its label inside the clip')
            labels[g] = np.where(labels[g] >= 0, labels[g], fl)
        src, dst = np.concatenate(src), np.concatenate(dst)
        clip = gt_masks[0].new_zeros((G * num_frames, H, W), dtype=torch.float32)
        if len(src):
            stack = torch.cat([m.to(torch.float32) for m in gt_masks])
            clip[torch.from_numpy(dst).to(dev)] = stack[torch.from_numpy(src).to(dev)]
        return clip.view(G, num_frames, H, W), torch.from_numpy(labels).to(dev), inst

    @staticmethod
    def tall(masks):
        """[F, Q, H, W] (a clip's frames) -> [Q, F*H, W]: the frames of a kernel stacked along the row axis (reference :147)."""
        F, Q, H, W = masks.shape
        return masks.permute(1, 0, 2, 3).reshape(Q, F * H, W)

    def assign(self, bbox_pred, cls_pred, gt_bboxes, gt_labels, gt_instance_ids, img_meta=None, gt_bboxes_ignore=None, eps=1e-7):
        """bbox_pred = mask logits of one clip [F, Q, H, W]; cls_pred [Q, ncls] | None -> (AssignResult, gt tall masks [G, F*H, W])."""
        assert gt_bboxes_ignore is None, 'Only case when gt_bboxes_ignore is None is supported.'
        F, Q, H, W = bbox_pred.shape
        clip, labels, _ = self.clip_instances(F, gt_bboxes, gt_labels, gt_instance_ids)
        gt_tall = clip.reshape(clip.shape[0], F * H, W)
        res = MaskHungarianAssigner.assign(self, self.tall(bbox_pred), cls_pred, gt_tall, labels, img_meta=img_meta)
        if gt_tall.shape[0] == 0:
            return res      # (the reference returns the bare AssignResult here as well, :137-142)
        return res, gt_tall


try:  # register beside the reference's class when mmdet is importable (same `type` name, force=True)
    from mmdet.core.bbox.builder import BBOX_ASSIGNERS  # type: ignore
    BBOX_ASSIGNERS.register_module(force=True)(MaskHungarianAssigner)
    BBOX_ASSIGNERS.register_module(force=True)(MaskHungarianAssignerVideo)
except Exception:  # noqa: BLE001
    pass

"""The loss tail of a training stage without the target tensors (include/vkn.h: "the loss tail of a training stage").

The reference's loop (knet/det/kernel_iter_head.py:139-231) runs, per stage and image, `assign -> sample -> get_targets -> loss`
(knet/det/mask_pseudo_sampler.py:15-205, knet/det/kernel_update_head.py:279-441): the sampler copies the matched ground-truth masks,
`get_targets` scatters them into a zero-filled [B (N + S), H, W] tensor and `loss` takes the positive rows out again.  `KernelIterHead.
_train_stages` takes this path instead when the assignment stayed on the device and the loss objects are the shipped ones:

  per STEP   `TailStep`: the batch's ground truth as ONE bank [G_total, H, W] (thing masks + stuff masks of every image; the assigner
             reads its cost operands from the same bank), label tensors as int64, a status word for range errors;
  per STAGE  `TailStep.stage_losses`: vkn_stage_targets (one launch: labels, label_weights, positive rows, bank row per row) and
             `StageTailFn` — focal pass, two mask-loss passes reading targets through the bank, one finalising workgroup (the five
             outputs incl. `pos_acc`), backward = one gradient pass over the mask logits + one scaling of the focal derivative.

Same values as `get_targets` + `loss` (tests/test_gpu_train.py: the reference goldens; tests/test_gpu_tail.py: against the op-by-op
path of this package).  Nothing here synchronises with the host: the number of positives is min(N, G_b) matched predictions plus the
present stuff classes per image — known from shapes."""
import ctypes

import torch

from . import _lib, ops
from ._lib import check
from .ops import _ptr, _req, _stream


def _is_sigmoid_focal(lc):
    """A sigmoid focal loss with mean reduction, checked BY VALUE: this package's `losses.FocalLoss` (with its fused kernel on) or
    mmdet's own `FocalLoss` — under real mmdet the registry keeps mmdet's class (registry.py), and the tail computes the same formula
    (mmdet/models/losses/focal_loss.py: py_sigmoid_focal_loss + weight_reduce_loss) from gamma / alpha / loss_weight."""
    from . import losses as L
    if type(lc) is L.FocalLoss:
        ours = True
    else:
        ours = False
        if type(lc).__name__ != 'FocalLoss' or not type(lc).__module__.startswith('mmdet.'):
            return False
    try:
        ok = (lc.use_sigmoid is True and lc.reduction == 'mean' and not getattr(lc, 'activated', False)
              and float(lc.gamma) >= 0.0 and 0.0 <= float(lc.alpha) <= 1.0 and float(lc.loss_weight) == float(lc.loss_weight))
    except (AttributeError, TypeError, ValueError):
        return False
    return ok and (lc.fused if ours else True)


def _is_sigmoid_dice(ld):
    """`DiceLoss(use_sigmoid=True, activate=True, reduction='mean')` — ours or mmdet's (not its later `naive_dice` variant)."""
    from . import losses as L
    if type(ld) is not L.DiceLoss and not (type(ld).__name__ == 'DiceLoss' and type(ld).__module__.startswith('mmdet.')):
        return False
    try:
        return bool(ld.use_sigmoid and ld.activate and ld.reduction == 'mean' and not getattr(ld, 'naive_dice', False)
                    and float(ld.eps) > 0.0 and float(ld.loss_weight) == float(ld.loss_weight))
    except (AttributeError, TypeError, ValueError):
        return False


def _shipped_losses(head):
    """the loss objects the fused tail restates: FocalLoss(sigmoid, mean) + the three mask losses of `_fused_mask_losses_ok`"""
    from . import losses as L
    lc, lm, ld, lr = head.loss_cls, head.loss_mask, head.loss_dice, head.loss_rank
    return (getattr(head, 'fused_mask_losses', False) and getattr(head, 'fused_tail', True)
            and _is_sigmoid_focal(lc)
            and type(lm) is L.CrossEntropyLoss and lm.use_sigmoid and lm.reduction == 'mean' and lm.class_weight is None
            and _is_sigmoid_dice(ld)
            and (lr is None or (type(lr) is L.CrossEntropyLoss and not lr.use_sigmoid and not lr.use_mask and lr.reduction == 'mean'
                                and lr.class_weight is None)))


class TailStep:
    """The ground truth of one training step, laid out for the fused tail.  `begin` returns None whenever a precondition fails — the
    caller then runs the op-by-op path (same values)."""

    @classmethod
    def begin(cls, iter_head, device, gt_masks, gt_labels, gt_sem_seg, gt_sem_cls):
        from .mask_hungarian_assigner import MaskHungarianAssigner
        from .mask_pseudo_sampler import MaskPseudoSampler
        if not getattr(iter_head, 'fused_tail', True) or device.type != 'cuda':
            return None
        if not all(type(a) is MaskHungarianAssigner and a.lsap == 'device' for a in iter_head.mask_assigner):
            return None
        if not all(type(s) is MaskPseudoSampler for s in iter_head.mask_sampler):
            return None
        if not all(_shipped_losses(h) for h in iter_head.mask_head):
            return None
        B = len(gt_masks)
        if B == 0 or len(gt_labels) != B:
            return None
        shape = tuple(gt_masks[0].shape[1:])
        for g, l in zip(gt_masks, gt_labels):
            if (not torch.is_tensor(g) or g.dim() != 3 or tuple(g.shape[1:]) != shape or not 0 < g.shape[0] <= 256 or g.device != device
                    or not torch.is_tensor(l) or l.numel() != g.shape[0]):
                return None
        with_sem = gt_sem_seg is not None and gt_sem_cls is not None and all(g is not None for g in gt_sem_seg) \
            and all(g is not None for g in gt_sem_cls)
        if with_sem:
            for g, c in zip(gt_sem_seg, gt_sem_cls):
                if not torch.is_tensor(g) or not torch.is_tensor(c) or c.numel() > 0 and (g.dim() != 3 or tuple(g.shape[1:]) != shape
                                                                                         or g.shape[0] != c.numel() or g.device != device):
                    return None
        if with_sem:
            # more stuff targets than stuff kernels cannot be distinct classes: the op-by-op path handles (and the reference defines) that
            n_stuff = max((getattr(h, 'num_stuff_classes', 0) for h in iter_head.mask_head), default=0)
            if any(c.numel() > n_stuff for c in gt_sem_cls):
                return None
        return cls(device, gt_masks, gt_labels, gt_sem_seg if with_sem else None, gt_sem_cls if with_sem else None)

    def __init__(self, device, gt_masks, gt_labels, gt_sem_seg, gt_sem_cls):
        self.device, self.B = device, len(gt_masks)
        self.with_sem = gt_sem_seg is not None
        parts, self.gt_row0, self.sem_row0, self.G, self.n_sem = [], [], [], [], []
        row = 0
        for b in range(self.B):
            self.gt_row0.append(row)
            self.G.append(int(gt_masks[b].shape[0]))
            parts.append(gt_masks[b])
            row += self.G[b]
            ns = int(gt_sem_cls[b].numel()) if self.with_sem else 0
            self.sem_row0.append(row)
            self.n_sem.append(ns)
            if ns:
                parts.append(gt_sem_seg[b])
                row += ns
        parts = [p if p.dtype == torch.float32 else p.float() for p in parts]
        self.bank = torch.cat(parts) if len(parts) > 1 else parts[0].contiguous()           # [G_total, H, W] fp32: ONE copy per step
        self.shape = tuple(self.bank.shape[1:])
        # what the assigner reads (cost kernels) — views of the bank: contiguous fp32, no per-stage conversion
        self.gt_views = [self.bank[self.gt_row0[b]:self.gt_row0[b] + self.G[b]] for b in range(self.B)]
        as_i64 = lambda t: t.to(device=device, dtype=torch.int64).contiguous().reshape(-1)  # noqa: E731  (no-op for device int64)
        self.labels = [as_i64(l) for l in gt_labels]
        self.sem_cls = [as_i64(c) if self.n_sem[b] else None for b, c in enumerate(gt_sem_cls)] if self.with_sem else [None] * self.B
        self.status = torch.zeros(1, dtype=torch.int32, device=device)    # bit 0: a stuff class out of range or listed twice (vkn_stage_targets), bit 1: a thing label (validate_labels)

    lowres_forward = True   # the forward sums of the low-res tail from the low-res logits (False: from their up-scaling, A/B)

    def stage_ok(self, head, assign_results, cls_score, scaled):
        return (cls_score is not None and cls_score.dtype == torch.float32 and scaled.is_cuda and scaled.dtype == torch.float32
                and scaled.dim() == 4 and tuple(scaled.shape[-2:]) == self.shape and (self.shape[0] * self.shape[1]) % 4 == 0
                and scaled.shape[0] == self.B and cls_score.shape[:2] == scaled.shape[:2]
                and len(assign_results) == self.B and all(getattr(r, '_pairs32', None) is not None for r in assign_results)
                and scaled.shape[1] > (head.num_stuff_classes if self.with_sem else 0))

    def stage_losses(self, head, cfg, assign_results, cls_score, scaled, lowres=None, stride=1):
        """-> dict(loss_cls, pos_acc, loss_mask, loss_dice[, loss_rank]) of one stage: the values of `head.loss(..., *head.get_targets(...))`.
        `lowres` (round 6): the stage's LOW-RES mask logits [B, Ns, h, w] under autograd, `scaled` their x`stride` up-scaling WITHOUT a graph
        (ops.upsample_bilinear of the detached logits): the losses are taken on `scaled`, the gradient goes straight to `lowres`
        (vkn_mask_losses_bwd_lowres_f32) — the up-scaled gradient tensor and the upsample's backward pass do not exist."""
        B, Ns = scaled.shape[:2]
        S, T = (head.num_stuff_classes, head.num_thing_classes) if self.with_sem else (0, 0)
        N, ncls = Ns - S, head.num_classes
        dev = self.device
        imgs = (_lib.VknTailImage * B)()
        pos0 = 0
        for b, r in enumerate(assign_results):
            rows, cols = r._pairs32
            k = int(rows.shape[0])
            imgs[b] = _lib.VknTailImage(rows.data_ptr(), cols.data_ptr(), self.labels[b].data_ptr(),
                                        self.sem_cls[b].data_ptr() if self.n_sem[b] else None, k, self.n_sem[b], self.gt_row0[b],
                                        self.sem_row0[b], pos0, 0)
            pos0 += k + self.n_sem[b]
        K, R = pos0, B * Ns
        if K == 0:
            return None
        t = _StageTargets()
        t.labels = torch.empty(R, dtype=torch.int64, device=dev)
        t.label_weights = torch.empty((R, ncls), dtype=torch.float32, device=dev)
        t.row_weight = torch.empty(R, dtype=torch.float32, device=dev)
        t.rowk = torch.empty(R, dtype=torch.int32, device=dev)
        t.tgt_row = torch.empty(R, dtype=torch.int32, device=dev)
        t.pos_rows = torch.empty(K, dtype=torch.int64, device=dev)
        pw = cfg['pos_weight'] if isinstance(cfg, dict) else cfg.pos_weight
        with torch.cuda.device(dev):
            check(_lib.lib().vkn_stage_targets(imgs, B, N, S, T, ncls, float(pw), t.labels.data_ptr(), _ptr(t.label_weights),
                                               _ptr(t.row_weight), t.rowk.data_ptr(), t.pos_rows.data_ptr(), t.tgt_row.data_ptr(),
                                               self.status.data_ptr(), _stream()))
        t.bank, t.K, t.B, t.Ns, t.ncls = self.bank, K, B, Ns, ncls
        from .losses import reduce_mean
        avg = reduce_mean(torch.tensor(float(K), device=dev)) if torch.distributed.is_available() and torch.distributed.is_initialized() else None
        t.avg_dev = avg.clamp_(min=1.0).reshape(1) if avg is not None else None     # (an all-reduced count lives on the device)
        t.avg_host = max(float(K), 1.0)
        lr = head.loss_rank
        t.cfg = _lib.VknTailCfg(float(head.loss_cls.loss_weight), float(head.loss_mask.loss_weight), float(head.loss_dice.loss_weight),
                                float(head.loss_dice.eps), float(lr.loss_weight) if lr is not None else 0.0, t.avg_host,
                                1 if lr is not None else 0)
        t.alpha, t.gamma = float(head.loss_cls.alpha), float(head.loss_cls.gamma)
        if lowres is not None:
            t.stride = int(stride)
            # forward sums from the low-res logits too (strides 2 / 4, up to 256 rows per frame); else from `scaled`
            from_low = self.lowres_forward and stride in (2, 4) and Ns <= 256 and lowres.shape[1] * lowres.shape[2] * lowres.shape[3] * 4 < 2 ** 31
            if not from_low and hasattr(scaled, 'materialize'):
                scaled = scaled.materialize()
            l_cls, acc, l_mask, l_dice, l_rank = StageTailFn.apply(cls_score, lowres, t, 'lowres' if from_low else scaled.detach())
        else:
            l_cls, acc, l_mask, l_dice, l_rank = StageTailFn.apply(cls_score, scaled, t, None)
        out = dict(loss_cls=l_cls, pos_acc=acc, loss_mask=l_mask, loss_dice=l_dice)
        if lr is not None:
            out['loss_rank'] = l_rank
        return out

    def finish(self):
        """hand the range-error word to the asynchronous flag queue (read without stalling: mask_hungarian_assigner.FLAGS)"""
        from .mask_hungarian_assigner import FLAGS
        FLAGS.push(self.status, 'gt_labels outside [0, num_thing_classes), or gt_sem_cls outside [num_thing_classes, num_classes) or listing a class twice')


class _StageTargets:
    """what vkn_stage_targets wrote for one stage (device tensors) + the constants of its losses"""


class StageTailFn(torch.autograd.Function):
    """(loss_cls, pos_acc [1], loss_mask, loss_dice, loss_rank) of one stage from (cls_score [B, Ns, ncls], scaled mask logits
    [B, Ns, H, W]) and the stage's targets `t`: four launches forward, two backward."""

    @staticmethod
    def forward(ctx, cls_score, mask_pred, t, scaled=None):
        # scaled is None: `mask_pred` is what the losses are taken on (and what the gradient is w.r.t.); else `mask_pred` are the low-res
        # logits (gradient target) and `scaled` their up-scaling (values only)
        L = _lib.lib()
        dev = mask_pred.device
        B, Ns, K, ncls = t.B, t.Ns, t.K, t.ncls
        # scaled == 'lowres' (round 6): `mask_pred` are the LOW-RES logits and the sums come straight from them
        # (vkn_mask_losses_fwd_lowres_f32) — their x`t.stride` up-scaling is not read, and need not exist
        from_low = isinstance(scaled, str)
        on = mask_pred if (scaled is None or from_low) else scaled
        R = B * Ns
        P = on.shape[2] * on.shape[3] * (t.stride ** 2 if from_low else 1)
        z = _req(cls_score.reshape(R, ncls), 'cls_score')
        pred = _req(on if from_low else on.reshape(R, P), 'mask_pred')
        with_rank = bool(t.cfg.with_rank)
        nbf = L.vkn_focal_loss_blocks(R, ncls)
        if from_low:
            nch = nbl = L.vkn_mask_losses_lowres_chunks(on.shape[2], on.shape[3])
        else:
            nch, nbl = L.vkn_mask_losses_chunks(P), L.vkn_mask_losses_blocks(P)
        part = torch.empty(nbf, dtype=torch.float32, device=dev)
        fgrad = torch.empty_like(z)
        rp = torch.empty((K, nch, 4), dtype=torch.float32, device=dev)
        lse = torch.empty((B, P), dtype=torch.float32, device=dev) if with_rank else None
        top = torch.empty((B, P), dtype=torch.int32, device=dev) if with_rank else None
        rkp = torch.empty((B, nbl), dtype=torch.float32, device=dev) if with_rank else None
        out = torch.empty(5, dtype=torch.float32, device=dev)
        a, bc = torch.empty(K, dtype=torch.float32, device=dev), torch.empty(K, dtype=torch.float32, device=dev)
        st = _stream()
        with torch.cuda.device(dev):
            check(L.vkn_focal_loss_f32(_ptr(z), t.labels.data_ptr(), _ptr(t.label_weights), 1 if ncls > 1 else 0, R, ncls, t.alpha, t.gamma,
                                       _ptr(part), _ptr(fgrad), st))
            if from_low:
                check(L.vkn_mask_losses_fwd_lowres_f32(_ptr(pred), _ptr(t.bank), t.tgt_row.data_ptr(), t.rowk.data_ptr(), K, B, Ns,
                                                       on.shape[2], on.shape[3], t.stride, 1 if with_rank else 0, _ptr(rp), _ptr(lse),
                                                       top.data_ptr() if with_rank else None, _ptr(rkp), st))
            else:
                check(L.vkn_mask_losses_fwd_bank_f32(_ptr(pred), _ptr(t.bank), t.tgt_row.data_ptr(), t.pos_rows.data_ptr(), t.rowk.data_ptr(),
                                                     K, B, Ns, P, 1 if with_rank else 0, _ptr(rp), _ptr(lse),
                                                     top.data_ptr() if with_rank else None, _ptr(rkp), st))
            check(L.vkn_stage_losses_final_f32(ctypes.byref(t.cfg), _ptr(t.avg_dev), _ptr(part), nbf, _ptr(rp), K, nch, _ptr(rkp),
                                               B * nbl if with_rank else 0, _ptr(z), t.labels.data_ptr(), t.pos_rows.data_ptr(), ncls, B, P,
                                               _ptr(out), _ptr(a), _ptr(bc), st))
        ctx.set_materialize_grads(False)
        ctx.lowres = scaled is not None
        ctx.save_for_backward(pred if from_low else (_req(mask_pred, 'mask_pred') if ctx.lowres else pred), fgrad, a, bc, lse if with_rank else a.new_empty(0),
                              top if with_rank else t.rowk.new_empty(0))
        ctx.t, ctx.shapes = t, (tuple(cls_score.shape), tuple(mask_pred.shape))
        acc = out[1:2]
        ctx.mark_non_differentiable(acc)
        return out[0], acc, out[2], out[3], out[4]

    @staticmethod
    def backward(ctx, g_cls, _g_acc, g_mask, g_dice, g_rank):
        pred, fgrad, a, bc, lse, top = ctx.saved_tensors
        t = ctx.t
        L = _lib.lib()
        with_rank = bool(t.cfg.with_rank)
        st = _stream()

        def scalar(g):
            if g is None:
                return None
            g = g.reshape(-1)
            return g if (g.dtype == torch.float32 and g.is_cuda) else g.to(device=pred.device, dtype=torch.float32)
        g_cls, g_mask, g_dice, g_rank = scalar(g_cls), scalar(g_mask), scalar(g_dice), scalar(g_rank)
        d_cls = d_pred = None
        with torch.cuda.device(pred.device):
            if g_cls is not None and ctx.needs_input_grad[0]:
                d_cls = torch.empty_like(fgrad)
                check(L.vkn_scale_by_f32(_ptr(fgrad), _ptr(g_cls), _ptr(t.avg_dev),
                                         float(t.cfg.w_cls) if t.avg_dev is not None else float(t.cfg.w_cls) / t.avg_host, _ptr(d_cls),
                                         fgrad.numel(), st))
                d_cls = d_cls.view(ctx.shapes[0])
            if ctx.needs_input_grad[1] and not (g_mask is None and g_dice is None and (g_rank is None or not with_rank)) and ctx.lowres:
                # the low-res logits: one pass, no up-scaled gradient tensor
                d_pred = torch.empty_like(pred)
                h, w = pred.shape[-2:]
                check(L.vkn_mask_losses_bwd_lowres_f32(_ptr(pred), _ptr(t.bank), t.tgt_row.data_ptr(), t.rowk.data_ptr(), _ptr(a), _ptr(bc),
                                                       _ptr(g_mask), _ptr(g_dice), _ptr(g_rank), float(t.cfg.w_mask), float(t.cfg.w_dice),
                                                       float(t.cfg.w_rank), t.K, _ptr(lse) if with_rank else None,
                                                       top.data_ptr() if with_rank else None, t.B, t.Ns, h, w, t.stride,
                                                       1 if with_rank else 0, _ptr(d_pred), st))
            elif ctx.needs_input_grad[1] and not (g_mask is None and g_dice is None and (g_rank is None or not with_rank)):
                R, P = pred.shape
                d_pred = torch.empty_like(pred)
                check(L.vkn_mask_losses_bwd_bank_f32(_ptr(pred), _ptr(t.bank), t.tgt_row.data_ptr(), t.rowk.data_ptr(), _ptr(a), _ptr(bc),
                                                     _ptr(g_mask), _ptr(g_dice), _ptr(g_rank), float(t.cfg.w_mask), float(t.cfg.w_dice),
                                                     float(t.cfg.w_rank), t.K, _ptr(lse) if with_rank else None,
                                                     top.data_ptr() if with_rank else None, t.B, t.Ns, P, 1 if with_rank else 0,
                                                     _ptr(d_pred), st))
                d_pred = d_pred.view(ctx.shapes[1])
        return d_cls, d_pred, None, None

"""`QuasiDenseMaskEmbedHeadGTMask` — drop-in for the embedding head between the kernel-update head's `object_feats_track` and the
quasi-dense tracker (knet/video/track_heads.py:552-718; call site knet/video/knet_quansi_dense_embed_fc_joint_train.py:613-623
`_track_forward`): same `HEADS` registration, ctor kwargs, module tree (`fcs.{i}.{weight,bias}`, `fc_embed.{weight,bias}`), method
names and returns.

At inference the head is an MLP on K rows (the thing segments a frame keeps for tracking): `relu(fc_i(x))` x num_fcs, then
`fc_embed` — one `vkn_linear_f32` launch per layer on pre-split (bf16x3) weights, inputs and outputs on the device, so the chain
head -> embeddings -> `vkn_qd_tracker_match_f32` has no host hop.  Under autograd (training) the same layers run as torch ops on the
module's parameters, like the [N x C] chain of the update head.  `num_convs > 0` (a 3x3 conv tower over RoI features) is not
provided: every shipped config sets `num_convs=0, roi_feat_size=1` (the "RoI feature" is a kernel).

The training side — `get_track_targets`, `match`, `loss` with `MultiPosCrossEntropyLoss` / `L2Loss`
(knet/video/qdtrack/losses/{multipos_cross_entropy_loss,l2_loss}.py) — is host-side torch on [K_key x K_ref] matrices.
"""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib, ops
from .losses import weight_reduce_loss
from .registry import HAVE_MM, build_loss, register_head


def cal_similarity(key_embeds, ref_embeds, method='dot_product', temperature=-1):
    """knet/video/qdtrack/track/similarity.py:5-25: [K_key, K_ref] dot products, cosine similarities, or cosine / temperature."""
    if method not in ('dot_product', 'cosine'):
        raise ValueError(method)
    if key_embeds.size(0) == 0 or ref_embeds.size(0) == 0:
        return torch.zeros((key_embeds.size(0), ref_embeds.size(0)), device=key_embeds.device)
    if method == 'cosine' or temperature > 0:
        sim = F.normalize(key_embeds, p=2, dim=1) @ F.normalize(ref_embeds, p=2, dim=1).t()
        return sim / temperature if (method == 'dot_product') else sim
    return key_embeds @ ref_embeds.t()


class MultiPosCrossEntropyLoss(nn.Module):
    """log(1 + sum_{p in pos} sum_{n in neg} exp(s_n - s_p)) per row (multipos_cross_entropy_loss.py:6-40), computed as
    softplus(logsumexp(neg) + logsumexp(-pos)) — the reference materialises the [K, R*R] difference matrix; same value."""

    def __init__(self, reduction='mean', loss_weight=1.0):
        super().__init__()
        self.reduction, self.loss_weight = reduction, loss_weight

    def forward(self, cls_score, label, weight=None, avg_factor=None, reduction_override=None, **kwargs):
        assert cls_score.size() == label.size()
        assert reduction_override in (None, 'none', 'mean', 'sum')
        reduction = reduction_override if reduction_override else self.reduction
        ninf = cls_score.new_full((), float('-inf'))
        lse_neg = torch.logsumexp(torch.where(label == 0, cls_score, ninf), dim=1)
        lse_pos = torch.logsumexp(torch.where(label == 1, -cls_score, ninf), dim=1)
        both = lse_neg + lse_pos                  # -inf when a row has no positive or no negative: loss log(1 + 0) = 0
        loss = torch.where(torch.isfinite(both), F.softplus(both), torch.zeros_like(both))
        if weight is not None:
            weight = weight.float()
        return self.loss_weight * weight_reduce_loss(loss, weight=weight, reduction=reduction, avg_factor=avg_factor)


class L2Loss(nn.Module):
    """|pred - target|^2 on the cosine similarities with margins and hard-negative mining (l2_loss.py:24-113)."""

    def __init__(self, neg_pos_ub=-1, pos_margin=-1, neg_margin=-1, hard_mining=False, reduction='mean', loss_weight=1.0):
        super().__init__()
        self.neg_pos_ub, self.pos_margin, self.neg_margin = neg_pos_ub, pos_margin, neg_margin
        self.hard_mining, self.reduction, self.loss_weight = hard_mining, reduction, loss_weight

    def forward(self, pred, target, weight=None, avg_factor=None, reduction_override=None):
        assert reduction_override in (None, 'none', 'mean', 'sum')
        reduction = reduction_override if reduction_override else self.reduction
        pred, weight, avg_factor = self.update_weight(pred, target, weight, avg_factor)
        assert pred.size() == target.size() and target.numel() > 0
        loss = torch.abs(pred - target) ** 2
        return self.loss_weight * weight_reduce_loss(loss, weight, reduction=reduction, avg_factor=avg_factor)

    def update_weight(self, pred, target, weight, avg_factor):
        """:74-113 — like the reference this writes -1 into `target` where the weight is not positive and shifts `pred` by the
        margins in place (the caller's tensors), then keeps at most neg_pos_ub negatives per positive: the hardest ones."""
        if weight is None:
            weight = target.new_ones(target.size())
        target[weight <= 0] = -1
        pos, neg = target == 1, target == 0
        if self.pos_margin > 0:
            pred[pos] -= self.pos_margin
        if self.neg_margin > 0:
            pred[neg] -= self.neg_margin
        pred = torch.clamp(pred, min=0, max=1)
        num_pos, num_neg = int(pos.sum()), int(neg.sum())
        if self.neg_pos_ub > 0 and num_neg / (num_pos + 1) > self.neg_pos_ub:
            keep = num_pos * self.neg_pos_ub
            idx = torch.nonzero(neg, as_tuple=False)
            if self.hard_mining:
                cost = (torch.abs(pred - target) ** 2)[idx[:, 0], idx[:, 1]].detach()
                idx = idx[cost.topk(keep)[1], :]
            else:
                idx = self.random_choice(idx, keep)
            chosen = torch.zeros_like(neg)
            chosen[idx[:, 0], idx[:, 1]] = True
            weight[torch.logical_xor(neg, chosen)] = 0
        return pred, weight, (weight > 0).sum()

    @staticmethod
    def random_choice(gallery, num):
        assert len(gallery) >= num
        if isinstance(gallery, list):
            gallery = np.array(gallery)
        cands = np.arange(len(gallery))
        np.random.shuffle(cands)
        pick = cands[:num]
        if not isinstance(gallery, np.ndarray):
            pick = torch.from_numpy(pick).long().to(gallery.device)
        return gallery[pick]


@register_head
class QuasiDenseMaskEmbedHeadGTMask(nn.Module):

    def __init__(self, num_convs=4, num_fcs=1, roi_feat_size=7, in_channels=256, conv_out_channels=256, fc_out_channels=1024,
                 embed_channels=256, conv_cfg=None, norm_cfg=None, softmax_temp=-1,
                 loss_track=dict(type='MultiPosCrossEntropyLoss', loss_weight=0.25),
                 loss_track_aux=dict(type='L2Loss', sample_ratio=3, margin=0.3, loss_weight=1.0, hard_mining=True)):
        super().__init__()
        if num_convs != 0:
            raise NotImplementedError('num_convs must be 0: every shipped config embeds kernels, not RoI feature maps '
                                      '(configs/det/video_knet_kitti_step/*_joint_train.py: num_convs=0, roi_feat_size=1)')
        self.num_convs, self.num_fcs, self.roi_feat_size = num_convs, num_fcs, roi_feat_size
        self.in_channels, self.conv_out_channels, self.fc_out_channels = in_channels, conv_out_channels, fc_out_channels
        self.embed_channels, self.conv_cfg, self.norm_cfg = embed_channels, conv_cfg, norm_cfg
        self.relu = nn.ReLU(inplace=True)
        self.convs = nn.ModuleList()
        self.fcs = nn.ModuleList()
        last = in_channels
        if num_fcs > 0:
            last *= roi_feat_size * roi_feat_size
            for i in range(num_fcs):
                self.fcs.append(nn.Linear(last if i == 0 else fc_out_channels, fc_out_channels))
            last = fc_out_channels
        self.fc_embed = nn.Linear(last, embed_channels)
        self.softmax_temp = softmax_temp
        self.loss_track = build_loss(loss_track)
        self.loss_track_aux = build_loss(loss_track_aux) if loss_track_aux is not None else None
        self._split = None        # pre-split (bf16x3) weights of the layers, keyed by the parameters' versions

    def init_weights(self):
        """:638-644"""
        for m in self.fcs:
            nn.init.xavier_uniform_(m.weight)
            nn.init.constant_(m.bias, 0)
        nn.init.normal_(self.fc_embed.weight, 0, 0.01)
        nn.init.constant_(self.fc_embed.bias, 0)

    # ---- inference: one HIP GEMM launch per layer, pre-split weights cached until a parameter changes
    def _layers(self):
        return list(self.fcs) + [self.fc_embed]

    def _split_weights(self, device):
        key = tuple((p.data_ptr(), p._version, str(p.device)) for m in self._layers() for p in (m.weight, m.bias))
        if self._split is None or self._split[0] != key:
            packs = []
            for m in self._layers():
                w = m.weight.detach()
                packs.append(ops.split_weight(w) if w.shape[1] % 32 == 0 else None)       # (K % 32 != 0: the exact-fp32 kernel)
            self._split = (key, packs)
        return self._split[1]

    def forward(self, x):
        """x [K, in_channels(, 1, 1)] -> embeddings [K, embed_channels]          (:646-656)"""
        if x.numel() == 0:       # a frame without thing segments (the reference's `view(0, -1)` cannot even express this)
            return x.new_zeros((0, self.embed_channels))
        x = x.reshape(x.size(0), -1)
        # the rule of KernelUpdateHead._needs_grad: grad mode on and anything differentiable in sight (input OR parameter) — eval() with
        # grad enabled and trainable parameters is differentiable too
        if torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in self.parameters())):
            if not x.is_cuda:
                raise _lib.VknLibraryError('QuasiDenseMaskEmbedHeadGTMask: expected CUDA/HIP tensors — the MI355X path has no CPU fallback')
            for fc in self.fcs:
                x = self.relu(fc(x))
            return self.fc_embed(x)
        if not x.is_cuda:
            raise _lib.VknLibraryError('QuasiDenseMaskEmbedHeadGTMask: expected CUDA/HIP tensors — the MI355X path has no CPU fallback')
        packs = self._split_weights(x.device)
        layers = self._layers()
        h = x.detach().float()
        for i, (m, ws) in enumerate(zip(layers, packs)):
            h = ops.linear(h, m.weight.detach(), m.bias.detach(), w_split=ws, act=1 if i + 1 < len(layers) else 0)
        return h

    # ---- training targets / similarities / losses (host-side torch on small matrices)
    def get_track_targets(self, gt_match_indices, key_sampling_results, ref_sampling_results):
        """:658-676 — targets[i][k, r] = 1 iff key positive k and reference positive r are the same instance; weight 1 for keys that
        have a partner."""
        track_targets, track_weights = [], []
        for match, key_res, ref_res in zip(gt_match_indices, key_sampling_results, ref_sampling_results):
            targets = match.new_zeros((key_res.pos_masks.size(0), ref_res.pos_masks.size(0)), dtype=torch.int)
            same = (match[key_res.pos_assigned_gt_inds].view(-1, 1) == ref_res.pos_assigned_gt_inds.view(1, -1)).int()
            targets[:, :same.size(1)] = same
            track_targets.append(targets)
            track_weights.append((targets.sum(dim=1) > 0).float())
        return track_targets, track_weights

    def match(self, key_embeds, ref_embeds, key_sampling_results, ref_sampling_results):
        """:678-697 — per image: dot-product (or cosine / temperature) similarities and, for the auxiliary loss, cosine ones."""
        key_embeds = torch.split(key_embeds, [res.pos_masks.size(0) for res in key_sampling_results])
        ref_embeds = torch.split(ref_embeds, [res.pos_masks.size(0) for res in ref_sampling_results])
        dists, cos_dists = [], []
        for k, r in zip(key_embeds, ref_embeds):
            dists.append(cal_similarity(k, r, method='dot_product', temperature=self.softmax_temp))
            cos_dists.append(cal_similarity(k, r, method='cosine') if self.loss_track_aux is not None else None)
        return dists, cos_dists

    def loss(self, dists, cos_dists, targets, weights):
        """:699-716"""
        losses = dict()
        loss_track, loss_track_aux = 0., 0.
        for d, c, t, w in zip(dists, cos_dists, targets, weights):
            loss_track = loss_track + self.loss_track(d, t, w, avg_factor=w.sum())
            if self.loss_track_aux is not None:
                loss_track_aux = loss_track_aux + self.loss_track_aux(c, t)
        losses['loss_track'] = loss_track / len(dists)
        if self.loss_track_aux is not None:
            losses['loss_track_aux'] = loss_track_aux / len(dists)
        return losses

    random_choice = staticmethod(L2Loss.random_choice)


def _register_losses():
    if HAVE_MM:       # with mmdet present: fill in what its LOSSES registry lacks (the reference's own qdtrack loss modules, if the
        from mmdet.models.builder import LOSSES as mm_losses  # type: ignore   # user imports them too, stay registered)
        for cls in (MultiPosCrossEntropyLoss, L2Loss):
            if mm_losses.get(cls.__name__) is None:
                mm_losses.register_module()(cls)
        return
    from .registry import LOSSES
    LOSSES.register_module(force=True)(MultiPosCrossEntropyLoss)
    LOSSES.register_module(force=True)(L2Loss)


_register_losses()

"""The YouTube-VIS model family's heads (BASELINE cfg4, `configs/video_knet_vis/_base_/models/knet_track_r50.py`) — SURVEY.md §8(f)-4:

* `KernelIterHeadVideo`       knet_vis/tracker/kernel_iter_head.py:13-330   per-frame roi head: S stages on bs * num_frames frames,
                                                                           instance-only results, `features` for the tracker
* `KernelUpdateHeadVideo`     knet_vis/tracker/kernel_update_head.py:20-380 clip-level stage: 5-D gather over the frames of a clip,
                                                                           kernels shared by the frames (`with_cls`) or per frame
* `KernelFrameIterHeadVideo`  knet_vis/tracker/kernel_frame_iter_head.py:15-383  the "tracker": query fusion + S clip-level stages

Same registry names, ctor kwargs and state-dict keys as the reference.  They are compositions of the same three HIP ops: the 5-D
gather `einsum('bfnhw,bfchw->bfnc')` is `vkn_mask_gather_f32` on B * F frames, the update runs through `vkn_stage_chain_f32`
(`query_merge_method='mean'`: x_feat averaged over the clip first) or the whole per-frame `vkn_stage_forward_f32`, the decode
`F.conv2d(mask_x[i], mask_feat[i])` with clip-shared kernels is `vkn_mask_decode_f32` with the kernels repeated per frame.
"""
import torch
import torch.nn as nn

from . import autograd as vag
from . import ops
from .kernel_iter_head import KernelIterHead
from .kernel_update_head import KernelUpdateHead, _FFNParams, _MHAParams
from .registry import BaseRoIHead, build_assigner, build_head, build_sampler, register_head


class _QueryMerge:
    """query_merge_method 'attention' / 'attention_pos' of both clip-level classes (tracker/kernel_update_head.py:153-179,
    tracker/kernel_frame_iter_head.py:47-75): the modules (same state-dict keys) and the one HIP entry point that runs them
    (`vkn_query_merge_f32`: 8 heads, FFN of 8 C, two LayerNorms)."""
    MERGE_NAMES = (None, 'query_merge_attn', 'query_merge_norm', 'query_merge_ffn', 'query_merge_ffn_norm')

    def _build_query_merge(self, channels):
        self.query_merge_attn = _MHAParams(channels, 8, 0.0)
        self.query_merge_norm = nn.LayerNorm(channels)
        self.query_merge_ffn = _FFNParams(channels, channels * 8, 2, 0.0)
        self.query_merge_ffn_norm = nn.LayerNorm(channels)
        self._merge_pack = None

    def invalidate_pack(self):
        self._merge_pack = None
        if hasattr(super(), 'invalidate_pack'):
            super().invalidate_pack()

    def _query_merge_autograd(self, query, keys, pos):
        """The same block as torch ops on this module's parameters (training): mmcv's MultiheadAttention adds the positions to
        query / key only; value and the residual are `keys` / `query` themselves."""
        F = keys.shape[1] // query.shape[1]
        q = query if pos is None else query + pos
        k = keys if pos is None else keys + pos.repeat(F, 1)
        att = self.query_merge_attn.attn(q.transpose(0, 1), k.transpose(0, 1), keys.transpose(0, 1), need_weights=False)[0]
        t = self.query_merge_norm(query + att.transpose(0, 1))
        return self.query_merge_ffn_norm(t + self.query_merge_ffn.layers(t))

    def _query_merge(self, query, keys, pos):
        """query [B,N,C], keys [B,F*N,C], pos [N,C] | None -> [B,N,C] on the GPU (no autograd)."""
        named = {k: v.detach() for k, v in self.named_parameters() if k.startswith('query_merge_')}
        sig = ops.StagePack.signature(named, query.device)
        if getattr(self, '_merge_pack', None) is None or self._merge_pack[0] != sig:
            self._merge_pack = (sig, ops.link_pack(named, query.device, *self.MERGE_NAMES))
        B, N, C = query.shape
        dims = ops.make_dims(B, N, C, 8, 8, 8, 8 * C, 1, 0, 0, 0.5, self.query_merge_norm.eps)
        return ops.query_merge(dims, self._merge_pack[1], query, keys, pos)


@register_head
class KernelUpdateHeadVideo(_QueryMerge, KernelUpdateHead):

    def __init__(self, with_cls=True, num_proposals=100, query_merge_method='mean', **kwargs):
        super().__init__(**kwargs)
        self.with_cls = with_cls
        self.num_proposals = num_proposals
        self.query_merge_method = query_merge_method
        if query_merge_method not in ('mean', 'attention', 'attention_pos'):
            raise NotImplementedError(query_merge_method)                                       # as the reference's forward (:264)
        if query_merge_method != 'mean' and with_cls:                                           # :155-179
            if self.conv_kernel_size != 1:
                raise NotImplementedError('Only supporting kernel size = 1')                    # :246
            self._build_query_merge(self.in_channels)
        if not with_cls:                       # the reference builds no classification branch then (:133-146): same state dict
            del self.cls_fcs
            del self.fc_cls
            self.num_cls_fcs = 0

    def init_weights(self):
        for p in self.parameters():
            if p.dim() > 1:
                nn.init.xavier_uniform_(p)
        if self.with_cls and self.loss_cls.use_sigmoid:
            nn.init.constant_(self.fc_cls.bias, -4.59511985013459)       # bias_init_with_prob(0.01) = -log(99)
        if self.kernel_init:
            nn.init.normal_(self.fc_mask.weight, mean=0, std=0.01)

    def make_dims(self, B, N, H, W):
        ncls = self.fc_cls.out_features if self.with_cls else 1
        return ops.make_dims(B, N, self.in_channels, H, W, self.num_heads, self.feedforward_channels, ncls, self.num_cls_fcs,
                             self.num_mask_fcs, self.hard_mask_thr, self.attention_norm.eps)

    def forward(self, x, proposal_feat, mask_preds, prev_cls_score=None, mask_shape=None, img_metas=None, pos=None):
        """x [B,F,C,H,W], mask_preds [B,F,N,H,W]; proposal_feat [B,N,C,1,1] (clip-level kernels, `with_cls` stages) or
        [B,F,N,C,1,1] (per-frame kernels) -> (cls_score | None, new_mask_preds [B,F,N,H,W], obj_feat)      reference :209-374"""
        B, F, C, H, W = x.shape
        if mask_preds.shape[-2:] != (H, W):      # reference :227-231: bilinear pre-resize of the incoming masks (no shipped config)
            N_ = mask_preds.shape[2]
            mask_preds = torch.nn.functional.interpolate(mask_preds.reshape(B * F, N_, *mask_preds.shape[-2:]), (H, W), mode='bilinear',
                                                         align_corners=False).reshape(B, F, N_, H, W)
        if self._needs_grad(x, proposal_feat):
            return self._forward_clip_autograd(x, proposal_feat, mask_preds, pos)
        if proposal_feat.dim() == 6:
            assert not self.with_cls
            N = proposal_feat.shape[2]
            assert self.num_proposals == N
            self._check_inputs(x.reshape(B * F, C, H, W), proposal_feat.reshape(B * F, N, C, 1, 1), mask_preds.reshape(B * F, N, H, W), None)
            _, masks, obj, _, _ = self._run(x.reshape(B * F, C, H, W), proposal_feat.reshape(B * F, N, C, 1, 1),
                                            mask_preds.reshape(B * F, N, H, W))
            return None, masks.reshape(B, F, N, H, W), obj.reshape(B, F, N, C, 1, 1)
        assert self.with_cls
        N = proposal_feat.shape[1]
        assert self.num_proposals == N
        xf, mf = x.reshape(B * F, C, H, W), mask_preds.reshape(B * F, N, H, W)
        self._check_inputs(xf, proposal_feat, mf, None)
        xraw, cnt = ops.mask_gather(xf, mf, self.hard_mask_thr)                                # einsum('bfnhw,bfchw->bfnc') :242
        if self.feat_transform is not None:                                                    # folded: x_feat = xraw W^T + cnt b
            w_ft = self.feat_transform.conv.weight.detach().reshape(C, C)
            x_feat = ops.linear(xraw.reshape(-1, C), w_ft).reshape(B * F, N, C)
            x_feat = x_feat + cnt.unsqueeze(-1) * self.feat_transform.conv.bias.detach()
        else:
            x_feat = xraw
        if self.query_merge_method == 'mean':
            x_feat = x_feat.reshape(B, F, N, C).mean(1)                                        # :243
        else:                                                                                  # :244-263
            if self.query_merge_method == 'attention_pos' and pos is None:
                raise ValueError("query_merge_method='attention_pos' needs `pos` (the tracker head's query_pos.weight)")
            x_feat = self._query_merge(proposal_feat.reshape(B, N, C), x_feat.reshape(B, F * N, C),
                                       pos.detach() if self.query_merge_method == 'attention_pos' else None)
        dims = self.make_dims(B, N, H, W)
        cls, kern, kb, obj = ops.stage_chain(dims, self.stage_pack(x.device), x_feat, proposal_feat.reshape(B, N, C))
        # every frame of a clip is decoded with the clip's kernels (:318-330)
        masks = ops.mask_decode(xf, kern.repeat_interleave(F, dim=0), kb.repeat_interleave(F, dim=0))
        return cls, masks.reshape(B, F, N, H, W), obj.reshape(B, N, C, 1, 1)

    def _forward_clip_autograd(self, x, proposal_feat, mask_preds, pos):
        """Training counterpart of `forward`: the gather / decode over the B * F frames are the HIP kernels behind their autograd
        Functions, the merge and the [N x C] chain torch ops on this module's parameters (base class `_xfeat_autograd` /
        `_chain_autograd`)."""
        B, F, C, H, W = x.shape
        xf = x.reshape(B * F, C, H, W)
        if proposal_feat.dim() == 6:                                                           # per-frame kernels (:266-267, :276-279)
            assert not self.with_cls
            N = proposal_feat.shape[2]
            assert self.num_proposals == N
            x_feat = self._xfeat_autograd(xf, mask_preds.reshape(B * F, N, H, W))
            _, kern, kb, obj, _ = self._chain_autograd(x_feat, proposal_feat.reshape(B * F, N, C, 1, 1))
            masks = vag.mask_decode(xf, kern, kb)
            return None, masks.reshape(B, F, N, H, W), obj.reshape(B, F, N, C, 1, 1)
        assert self.with_cls
        N = proposal_feat.shape[1]
        assert self.num_proposals == N
        x_feat = self._xfeat_autograd(xf, mask_preds.reshape(B * F, N, H, W)).reshape(B, F, N, C)
        if self.query_merge_method == 'mean':
            x_feat = x_feat.mean(1)                                                            # :243
        else:                                                                                  # :244-263 (init_query is detached there)
            if self.query_merge_method == 'attention_pos' and pos is None:
                raise ValueError("query_merge_method='attention_pos' needs `pos` (the tracker head's query_pos.weight)")
            x_feat = self._query_merge_autograd(proposal_feat.reshape(B, N, C).detach(), x_feat.reshape(B, F * N, C),
                                                pos if self.query_merge_method == 'attention_pos' else None)
        cls, kern, kb, obj, _ = self._chain_autograd(x_feat, proposal_feat.reshape(B, N, C, 1, 1))
        masks = vag.mask_decode(xf, kern.repeat_interleave(F, dim=0), kb.repeat_interleave(F, dim=0) if kb is not None else None)
        return cls, masks.reshape(B, F, N, H, W), obj

    def get_targets(self, sampling_results, rcnn_train_cfg, concat=True, gt_sem_seg=None, gt_sem_cls=None):
        """knet_vis's signature (tracker/kernel_update_head.py:504-530): no gt_mask / gt_labels arguments."""
        return KernelUpdateHead.get_targets(self, sampling_results, None, None, rcnn_train_cfg, concat, gt_sem_seg, gt_sem_cls)

    def get_seg_masks_tracking(self, masks_per_img, labels_per_img, scores_per_img, ids_per_img, test_cfg, img_meta):
        """knet_vis/det/kernel_update_head.py:484-500: masks + mmtrack's `outs2results` packing (ids instead of boxes)."""
        seg_masks = self.rescale_masks(masks_per_img, img_meta) > self._meta(test_cfg, 'mask_thr')
        bboxes = torch.zeros((masks_per_img.shape[0], 5), dtype=torch.float32)
        bboxes[:, -1] = scores_per_img.cpu()
        tracks = outs2results(bboxes=bboxes, labels=labels_per_img, masks=seg_masks, ids=ids_per_img, num_classes=self.num_classes)
        return tracks['bbox_results'], tracks['mask_results']


KernelUpdateHead.get_seg_masks_tracking = KernelUpdateHeadVideo.get_seg_masks_tracking   # the per-frame head has it too (:484)


def outs2results(bboxes=None, labels=None, masks=None, ids=None, num_classes=None, **kwargs):
    """mmtrack.core `outs2results` (third-party, mmtrack 0.x; restated): per-class lists; with `ids` the box rows are
    [id, x1, y1, x2, y2, score] and detections with id < 0 are dropped."""
    import numpy as np
    assert labels is not None and num_classes is not None
    results = dict()
    labels = labels.cpu().numpy() if torch.is_tensor(labels) else np.asarray(labels)
    if ids is not None:
        ids = ids.cpu().numpy() if torch.is_tensor(ids) else np.asarray(ids)
        valid = ids > -1
        ids, labels = ids[valid], labels[valid]
    else:
        valid = np.ones(len(labels), dtype=bool)
    if bboxes is not None:
        bb = (bboxes.cpu().numpy() if torch.is_tensor(bboxes) else np.asarray(bboxes))[valid]
        if ids is None:
            results['bbox_results'] = [bb[labels == i, :] for i in range(num_classes)]
        else:
            rows = np.concatenate([ids[:, None].astype(bb.dtype), bb], axis=1) if bb.shape[0] else np.zeros((0, 6), dtype=np.float32)
            results['bbox_results'] = [rows[labels == i, :] for i in range(num_classes)]
    if masks is not None:
        mm = (masks.cpu().numpy() if torch.is_tensor(masks) else np.asarray(masks))[valid]
        mask_results = [[] for _ in range(num_classes)]
        for i in range(mm.shape[0]):
            mask_results[labels[i]].append(mm[i])
        results['mask_results'] = mask_results
    return results


@register_head
class KernelIterHeadVideo(KernelIterHead):
    """knet_vis/tracker/kernel_iter_head.py: the per-frame roi head of the VIS model.  x holds bs * num_frames frames."""

    def init_assigner_sampler(self):
        super().init_assigner_sampler()
        for a in self.mask_assigner:       # knet_vis's DiceCost / MaskCost do not clamp the sigmoid (knet_vis/det/mask_hungarian_assigner.py:69,100)
            if hasattr(a, 'pred_clamp'):
                a.pred_clamp = (0.0, 0.0)

    def forward_train(self, x, proposal_feats, mask_preds, cls_score, ref_img_metas, gt_masks, gt_labels, gt_bboxes_ignore=None,
                      imgs_whwh=None, gt_bboxes=None, gt_sem_seg=None, gt_sem_cls=None):
        """-> (losses, features) over the bs * num_frames frames (reference :139-242): per-frame ground truth arrives as
        `gt_masks[clip][frame]` and `gt_labels[clip]` rows (frame, label) and is flattened to the frame list the image head's stage
        loop takes; `features` are the clip-shaped views the clip-level tracker head reads."""
        num_imgs, num_frames = len(ref_img_metas), len(ref_img_metas[0])
        metas = [m for clip in ref_img_metas for m in clip]
        flat_masks, flat_labels = [], []
        for i in range(num_imgs):
            rows = gt_labels[i]
            for j in range(num_frames):
                flat_masks.append(gt_masks[i][j])
                flat_labels.append(rows[:, 1][rows[:, 0] == j])
        losses, last = self._train_stages(x, proposal_feats, mask_preds, cls_score, metas, flat_masks, flat_labels, imgs_whwh=imgs_whwh,
                                          gt_sem_seg=gt_sem_seg, gt_sem_cls=gt_sem_cls)
        object_feats, cls_last, masks_last = last['object_feats'], last['cls_score'], last['mask_preds']
        nq, c = object_feats.shape[1:3]
        h, w = x.shape[-2:]
        assert object_feats.shape[0] == x.shape[0] == num_imgs * num_frames and c == x.shape[1]
        features = dict(obj_feats=object_feats.reshape((num_imgs, num_frames) + tuple(object_feats.shape[1:])),
                        x_feats=x.reshape(num_imgs, num_frames, c, h, w),
                        cls_scores=cls_last.reshape(num_imgs, num_frames, nq, self.num_classes),
                        masks=masks_last.reshape(num_imgs, num_frames, nq, h, w))
        return losses, features

    def merge_stuff_thing(self, total_masks, total_labels, total_scores, merge_cfg=None):
        """The joint merge of SOFT masks with the VIS encoding (reference :352-388): a pixel carries `label + segment * 1000`
        (mmdet's INSTANCE_OFFSET; segments counted from 0 in descending score order), `num_classes` where nothing survives.
        -> int64 ndarray [H, W]."""
        winner, seg_id, _ = self._joint_merge(total_masks, total_labels, total_scores, merge_cfg)
        code = torch.where(seg_id > 0, total_labels.long() + (seg_id - 1) * 1000, seg_id.new_full((), self.num_classes))
        return code[winner].cpu().numpy()

    def simple_test(self, x, proposal_feats, mask_preds, cls_score, img_metas, ref_img_metas, imgs_whwh=None, rescale=False):
        """-> (results: per frame `(bbox_result, segm_result)`, features: the tracker's inputs)            reference :243-313"""
        if self.do_panoptic:
            raise NotImplementedError  # as the reference (:287-288)
        if not self._fused_ok(x):
            raise NotImplementedError('simple_test needs the fused GPU head (eval mode, CUDA tensors)')
        num_imgs, num_frames = len(ref_img_metas), len(ref_img_metas[0])
        obj, cls, masks, scaled, _ = self._head_forward(x, proposal_feats, mask_preds)
        bs_nf, nq, c, k1, k2 = obj.shape
        h, w = x.shape[-2:]
        assert bs_nf == num_imgs * num_frames == x.shape[0] and c == x.shape[1]
        features = dict(obj_feats=obj.reshape(num_imgs, num_frames, nq, c, k1, k2), x_feats=x.reshape(num_imgs, num_frames, c, h, w),
                        cls_scores=cls.reshape(num_imgs, num_frames, nq, self.num_classes),
                        masks=masks.reshape(num_imgs, num_frames, nq, h, w))
        results = []
        for img_id in range(num_imgs):
            for frame_id in range(num_frames):
                i = img_id * num_frames + frame_id
                results.append(self._instance_result(cls[i], scaled[i], img_metas[img_id]))
        return results, features


@register_head
class KernelFrameIterHeadVideo(_QueryMerge, BaseRoIHead):
    """knet_vis/tracker/kernel_frame_iter_head.py:15-383 (inference; every `query_merge_method`, `with_mask_init` optional)."""

    def __init__(self, mask_head=None, with_mask_init=False, num_stages=3, stage_loss_weights=(1, 1, 1), proposal_feature_channel=256,
                 assign_stages=5, num_proposals=100, num_thing_classes=80, num_stuff_classes=53, query_merge_method='mean',
                 train_cfg=None, test_cfg=None, pretrained=None, init_cfg=None, **kwargs):
        assert len(stage_loss_weights) == num_stages
        self.num_stages = num_stages
        self.stage_loss_weights = stage_loss_weights
        self.assign_stages = assign_stages
        self.num_proposals = num_proposals
        self.num_thing_classes = num_thing_classes
        self.num_stuff_classes = num_stuff_classes
        self.query_merge_method = query_merge_method
        self.proposal_feature_channel = proposal_feature_channel
        if query_merge_method not in ('mean', 'attention', 'attention_pos'):
            raise NotImplementedError(query_merge_method)
        super().__init__(mask_head=mask_head, train_cfg=train_cfg, test_cfg=test_cfg, **kwargs)
        if query_merge_method != 'mean':                                                        # :47-75
            self.init_query = nn.Embedding(num_proposals, proposal_feature_channel)
            if query_merge_method == 'attention_pos':
                self.query_pos = nn.Embedding(num_proposals, proposal_feature_channel)
            self._build_query_merge(proposal_feature_channel)
        self.with_mask_init = with_mask_init
        if self.with_mask_init:
            self.fc_mask = nn.Linear(proposal_feature_channel, proposal_feature_channel)

    def init_mask_head(self, bbox_roi_extractor=None, mask_head=None):
        assert bbox_roi_extractor is None
        self.mask_head = nn.ModuleList()
        if not isinstance(mask_head, list):
            mask_head = [mask_head for _ in range(self.num_stages)]
        assert len(mask_head) == self.num_stages
        for idx, head in enumerate(mask_head):
            head = dict(head)
            head.update(with_cls=(idx < self.assign_stages))                                   # reference :100
            self.mask_head.append(build_head(head))

    def init_assigner_sampler(self):
        """One assigner + sampler per stage, all from the ONE `train_cfg.tracker` dict (:92-103)."""
        self.mask_assigner, self.mask_sampler = [], []
        if self.train_cfg is not None:
            for i in range(self.num_stages):
                self.mask_assigner.append(build_assigner(self._cfg(self.train_cfg, 'assigner')))
                self.current_stage = i
                self.mask_sampler.append(build_sampler(self._cfg(self.train_cfg, 'sampler'), context=self))

    def init_bbox_head(self, mask_roi_extractor, mask_head):
        raise NotImplementedError

    def init_weights(self):
        for h in self.mask_head:
            h.init_weights()

    def _mask_forward(self, stage, x, object_feats, mask_preds):
        mask_head = self.mask_head[stage]
        pos = self.query_pos.weight if self.query_merge_method == 'attention_pos' else None     # :117
        cls_score, mask_preds, object_feats = mask_head(x, object_feats, mask_preds, img_metas=None, pos=pos)
        if mask_head.mask_upsample_stride > 1 and (stage == self.num_stages - 1 or self.training):
            scaled = self._upsample_clip(mask_preds, mask_head.mask_upsample_stride)                              # :121-130
        else:
            scaled = mask_preds
        return dict(cls_score=cls_score, mask_preds=mask_preds, scaled_mask_preds=scaled, object_feats=object_feats)

    @staticmethod
    def _upsample_clip(mask_preds, s):
        """[B,F,N,H,W] -> [B,F,N,sH,sW] bilinear: the HIP kernel at inference, torch's op (for its backward) under autograd."""
        B, F, N, H, W = mask_preds.shape
        flat = mask_preds.reshape(B * F, N, H, W)
        if mask_preds.requires_grad and torch.is_grad_enabled():
            up = vag.upsample_bilinear(flat.contiguous(), s)
        else:
            up = ops.upsample_bilinear(flat, s)
        return up.reshape(B, F, N, H * s, W * s)

    def _query_fusion(self, obj_feats, num_imgs, num_frames):
        if self.query_merge_method == 'mean':
            return obj_feats.mean(1)                                                           # :140-141
        if tuple(obj_feats.shape[-2:]) != (1, 1):
            raise NotImplementedError('Only supporting kernel size = 1')                       # :143
        C, N = self.proposal_feature_channel, self.num_proposals                                # :142-160
        keys = obj_feats.reshape(num_imgs, num_frames * N, C)
        if torch.is_grad_enabled() and (keys.requires_grad or self.init_query.weight.requires_grad):
            pos = self.query_pos.weight if self.query_merge_method == 'attention_pos' else None
            return self._query_merge_autograd(self.init_query.weight.expand(num_imgs, N, C), keys, pos)[..., None, None]
        query = self.init_query.weight.detach().expand(num_imgs, N, C).contiguous()
        pos = self.query_pos.weight.detach() if self.query_merge_method == 'attention_pos' else None
        return self._query_merge(query, keys, pos)[..., None, None]

    def _mask_init(self, object_feats, x_feats, num_imgs):
        """:163-178: mask_preds = conv2d(x_feats[i], fc_mask(object_feats)[i]) for all frames of clip i."""
        B, F, C, H, W = x_feats.shape
        N = object_feats.shape[1]
        if torch.is_grad_enabled() and (object_feats.requires_grad or x_feats.requires_grad or self.fc_mask.weight.requires_grad):
            k = self.fc_mask(object_feats.reshape(B, N, C))
            return vag.mask_decode(x_feats.reshape(B * F, C, H, W), k.repeat_interleave(F, dim=0), None).reshape(B, F, N, H, W)
        k = ops.linear(object_feats.reshape(B * N, C), self.fc_mask.weight.detach(), self.fc_mask.bias.detach()).reshape(B, N, C)
        return ops.mask_decode(x_feats.reshape(B * F, C, H, W), k.repeat_interleave(F, dim=0)).reshape(B, F, N, H, W)

    @staticmethod
    def _cfg(cfg, key):
        return cfg[key] if isinstance(cfg, dict) else getattr(cfg, key)

    def _clip_losses(self, stage, prefix, object_feats, cls_score, scaled, assigned, ref_gt_masks, ref_gt_labels, ref_gt_instance_ids,
                     cls_for_assign, out):
        """Assign (or re-use `assigned`, the (AssignResult, gt tall masks) pairs of the last assigning stage), sample, build the
        targets and add this stage's losses to `out` under `prefix`.  A clip's masks are handled in the "tall" layout
        ([Q, F*H, W]: frames stacked along the rows) — losses and costs then are the per-frame ones, unchanged.  Returns `assigned`."""
        assigner, sampler, head = self.mask_assigner[stage], self.mask_sampler[stage], self.mask_head[stage]
        num_imgs = scaled.shape[0]
        if assigned is None:
            assigned = []
            det = scaled.detach()
            for i in range(num_imgs):
                r = assigner.assign(det[i][:, :self.num_proposals], cls_for_assign[i] if cls_for_assign is not None else None,
                                    ref_gt_masks[i], ref_gt_labels[i], ref_gt_instance_ids[i])
                if not isinstance(r, tuple):   # a clip without ground truth
                    F, _, H, W = det[i].shape
                    r = (r, det.new_zeros((0, F * H, W)))
                assigned.append(r)
        tall = torch.stack([assigner.tall(scaled[i]) for i in range(num_imgs)])                          # [B, Q, F*H, W]
        sampling = [sampler.sample(assigned[i][0], tall[i], assigned[i][1]) for i in range(num_imgs)]
        targets = head.get_targets(sampling, self.train_cfg, True, gt_sem_seg=None, gt_sem_cls=None)
        for key, value in head.loss(object_feats, cls_score, tall, *targets).items():
            out[f'{prefix}_{key}'] = value * self.stage_loss_weights[stage]
        return assigned

    def forward_train(self, x, ref_img_metas, cls_scores, masks, obj_feats, ref_gt_masks, ref_gt_labels, ref_gt_instance_ids, **kwargs):
        """Clip-level training of the tracker head (reference :182-312): x [B,F,C,H,W]; ref_gt_masks[i][f] = [n_f, sH, sW] masks of
        frame f of clip i, ref_gt_labels[i] / ref_gt_instance_ids[i] = [M, 2] rows (frame, label) / (frame, instance id).
        -> (losses `tracker_s{stage}_*` (+ `tracker_init_*` with `with_mask_init`), features).  Stages < assign_stages assign on their
        own predictions; the per-frame stages after them re-use the last assignment (:270-283)."""
        if not self.mask_assigner:
            raise RuntimeError('forward_train needs train_cfg (assigner / sampler / pos_weight of the tracker head)')
        num_imgs, num_frames = len(ref_img_metas), len(ref_img_metas[0])
        object_feats = self._query_fusion(obj_feats, num_imgs, num_frames) if obj_feats.dim() == 6 else obj_feats
        losses = {}
        if self.with_mask_init:                                                                          # :196-246
            mask_preds = self._mask_init(object_feats, x, num_imgs)
            s = self.mask_head[0].mask_upsample_stride
            scaled = self._upsample_clip(mask_preds, s) if s > 1 else mask_preds
            self._clip_losses(0, 'tracker_init', object_feats, None, scaled, None, ref_gt_masks, ref_gt_labels, ref_gt_instance_ids,
                              None, losses)
        else:
            mask_preds = masks
        assigned, cls_score = None, None
        for stage in range(self.num_stages):
            if stage == self.assign_stages:
                object_feats = object_feats[:, None].repeat(1, num_frames, 1, 1, 1, 1)
            r = self._mask_forward(stage, x, object_feats, mask_preds)
            mask_preds, cls_score, object_feats = r['mask_preds'], r['cls_score'], r['object_feats']
            cls_for_assign = None
            if stage < self.assign_stages:
                assigned = None
                if cls_score is not None:
                    cls_for_assign = cls_score.detach()[:, :self.num_proposals, :self.num_thing_classes]
            assigned = self._clip_losses(stage, f'tracker_s{stage}', object_feats, cls_score, r['scaled_mask_preds'], assigned,
                                         ref_gt_masks, ref_gt_labels, ref_gt_instance_ids, cls_for_assign, losses)
        self.mask_assigner[0].check_status(*self.mask_assigner[1:], wait=False)
        return losses, dict(obj_feats=object_feats, x_feats=x, cls_scores=cls_score, masks=mask_preds)

    def simple_test(self, x, img_metas, ref_img_metas, cls_scores, masks, obj_feats, **kwargs):
        """-> (results[img][frame] = (bbox_results with ids, mask_results), features)                      reference :313-372"""
        num_imgs, num_frames = len(ref_img_metas), len(ref_img_metas[0])
        object_feats = self._query_fusion(obj_feats, num_imgs, num_frames) if obj_feats.dim() == 6 else obj_feats
        mask_preds = self._mask_init(object_feats, x, num_imgs) if self.with_mask_init else masks
        cls_score = None
        with torch.no_grad():
            for stage in range(self.num_stages):
                if stage == self.assign_stages:
                    object_feats = object_feats[:, None].repeat(1, num_frames, 1, 1, 1, 1)
                r = self._mask_forward(stage, x, object_feats, mask_preds)
                mask_preds, scaled_mask_preds = r['mask_preds'], r['scaled_mask_preds']
                cls_score = r['cls_score'] if r['cls_score'] is not None else cls_score
                object_feats = r['object_feats']
        last = self.mask_head[-1]
        num_classes = last.num_classes
        cls_score = cls_score.sigmoid() if last.loss_cls.use_sigmoid else cls_score.softmax(-1)[..., :-1]
        k = self._cfg(self.test_cfg, 'max_per_img')
        results = []
        for img_id in range(num_imgs):
            scores, topk = cls_score[img_id].flatten(0, 1).topk(k, sorted=True)
            mask_indices, labels = topk // num_classes, topk % num_classes
            results.append([last.get_seg_masks_tracking(scaled_mask_preds[img_id][f][mask_indices], labels, scores, torch.arange(k),
                                                        self.test_cfg, img_metas[img_id]) for f in range(num_frames)])
        features = dict(obj_feats=object_feats, x_feats=x, cls_scores=cls_score, masks=mask_preds)
        return results, features

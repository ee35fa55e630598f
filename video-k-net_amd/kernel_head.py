"""`ConvKernelHead` — drop-in for the reference's kernel-initialisation head ("RPN" of K-Net,
knet/det/kernel_head.py:12-263, 506-515): same `HEADS` registration, ctor kwargs, state-dict keys (`init_kernels.weight`,
`conv_seg.{weight,bias}`, `loc_convs.{i}.{conv,gn}.*`, `seg_convs.{i}.*`, `ins_downsample.*`, `seg_downsample.*`) and returns.

Scope: everything from the localization FPN's two feature maps to the first `KernelUpdateHead` —
`init_kernels` / `conv_seg` 1x1 convs, `x_feats = semantic + loc`, the object-feature gather and the stuff-kernel concatenation —
is ONE C-ABI call (`vkn_kernel_init_f32`: decode with frame-shared kernels + gather, the same HIP kernels as the head).
The layers UPSTREAM of that (`localization_fpn`, the GroupNorm'ed `loc_convs` / `seg_convs` / `*_downsample`) belong to the backbone
side, out of the hot-path scope: they are kept as ordinary torch modules so that checkpoints load and the class is usable
end to end, and run on the GPU through PyTorch-ROCm.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import autograd as vag
from . import ops
from .losses import accuracy, reduce_mean
from .registry import HAVE_MM, HEADS, build_assigner, build_loss, build_sampler, register_head


class _ConvGNReLU(nn.Module):
    """mmcv `ConvModule(cin, cout, k, stride, padding, norm_cfg=GN)`: conv (no bias, a norm follows) -> GroupNorm -> ReLU, with
    mmcv's attribute names (`conv`, `gn`) so the state-dict keys match.  Upstream of the hot path (see module docstring)."""

    def __init__(self, cin, cout, k, stride=1, padding=0, norm_cfg=None):
        super().__init__()
        norm_cfg = dict(norm_cfg or dict(type='GN', num_groups=32))
        if norm_cfg.get('type') != 'GN':
            raise NotImplementedError('ConvKernelHead norm_cfg must be GN (every shipped config)')
        self.conv = nn.Conv2d(cin, cout, k, stride=stride, padding=padding, bias=False)
        self.gn = nn.GroupNorm(norm_cfg.get('num_groups', 32), cout)
        self.activate = nn.ReLU(inplace=True)

    def forward(self, x):
        return self.activate(self.gn(self.conv(x)))


@register_head
class ConvKernelHead(nn.Module):

    def __init__(self, num_proposals=100, in_channels=256, out_channels=256, num_heads=8, num_cls_fcs=1, num_seg_convs=1,
                 num_loc_convs=1, att_dropout=False, localization_fpn=None, conv_kernel_size=1,
                 norm_cfg=dict(type='GN', num_groups=32), semantic_fpn=True, train_cfg=None, num_classes=80,
                 xavier_init_kernel=False, kernel_init_std=0.01, use_binary=False, proposal_feats_with_obj=False,
                 loss_mask=None, loss_seg=None, loss_cls=None, loss_dice=None, loss_rank=None, feat_downsample_stride=1,
                 feat_refine_stride=1, feat_refine=True, with_embed=False, feat_embed_only=False, conv_normal_init=False,
                 mask_out_stride=4, hard_target=False, num_thing_classes=80, num_stuff_classes=53, mask_assign_stride=4,
                 ignore_label=255, thing_label_in_seg=0, cat_stuff_mask=False, **kwargs):
        super().__init__()
        if conv_kernel_size != 1:
            raise NotImplementedError('conv_kernel_size must be 1 (every shipped config)')
        if cat_stuff_mask and not semantic_fpn:
            raise ValueError('cat_stuff_mask needs semantic_fpn=True')
        self.num_proposals = num_proposals
        self.num_cls_fcs = num_cls_fcs
        self.train_cfg = train_cfg
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.num_classes = num_classes
        self.proposal_feats_with_obj = proposal_feats_with_obj
        self.sampling = False
        self.semantic_fpn = semantic_fpn
        self.norm_cfg = norm_cfg
        self.num_heads = num_heads
        self.att_dropout = att_dropout
        self.mask_out_stride = mask_out_stride
        self.hard_target = hard_target
        self.conv_kernel_size = conv_kernel_size
        self.xavier_init_kernel = xavier_init_kernel
        self.kernel_init_std = kernel_init_std
        self.feat_downsample_stride = feat_downsample_stride
        self.feat_refine_stride = feat_refine_stride
        self.conv_normal_init = conv_normal_init
        self.feat_refine = feat_refine
        self.with_embed = with_embed
        self.feat_embed_only = feat_embed_only
        self.num_loc_convs = num_loc_convs
        self.num_seg_convs = num_seg_convs
        self.use_binary = use_binary
        self.num_thing_classes = num_thing_classes
        self.num_stuff_classes = num_stuff_classes
        self.mask_assign_stride = mask_assign_stride
        self.ignore_label = ignore_label
        self.thing_label_in_seg = thing_label_in_seg
        self.cat_stuff_mask = cat_stuff_mask
        self.loss_cfgs = dict(loss_mask=loss_mask, loss_seg=loss_seg, loss_cls=loss_cls, loss_dice=loss_dice, loss_rank=loss_rank)
        self.localization_fpn = self._build_neck(localization_fpn)
        # losses, assigner and sampler exactly as the reference builds them (knet/det/kernel_head.py:90-120)
        for key, cfg in self.loss_cfgs.items():
            setattr(self, key, build_loss(dict(cfg)) if cfg is not None else None)
        if self.train_cfg:
            self.assigner = build_assigner(self._cfg(self.train_cfg, 'assigner'))
            self.sampler = build_sampler(dict(type='MaskPseudoSampler'), context=self)     # `self.sampling` is False (:56, :113-117)
        self._init_layers()

    @staticmethod
    def _cfg(cfg, key):
        return cfg[key] if isinstance(cfg, dict) else getattr(cfg, key)

    @staticmethod
    def _build_neck(cfg):
        """`build_neck(localization_fpn)` (knet/det/kernel_head.py:63): an nn.Module is taken as is; a config dict is built
        through mmdet's registry when mmdet is importable, else through the bundled one (where a user registers their neck)."""
        if cfg is None or isinstance(cfg, nn.Module):
            return cfg
        if HAVE_MM:
            from mmdet.models.builder import build_neck  # type: ignore
            return build_neck(cfg)
        if cfg.get('type') in HEADS:
            return HEADS.build(cfg)
        raise NotImplementedError(f'localization_fpn type {cfg.get("type")!r} is not registered (the FPN neck is backbone-side, '
                                  'outside this package: pass an nn.Module or register one in video_k_net_amd.HEADS)')

    def _init_layers(self):
        """knet/det/kernel_head.py:122-168 — same attribute names, creation order and shapes."""
        self.init_kernels = nn.Conv2d(self.out_channels, self.num_proposals, 1, padding=0, bias=False)
        if self.semantic_fpn:
            self.conv_seg = nn.Conv2d(self.out_channels, self.num_classes, 1)
        if self.feat_downsample_stride > 1 and self.feat_refine:
            self.ins_downsample = _ConvGNReLU(self.in_channels, self.out_channels, 3, self.feat_refine_stride, 1, self.norm_cfg)
            self.seg_downsample = _ConvGNReLU(self.in_channels, self.out_channels, 3, self.feat_refine_stride, 1, self.norm_cfg)
        self.loc_convs = nn.ModuleList(
            [_ConvGNReLU(self.in_channels, self.out_channels, 1, norm_cfg=self.norm_cfg) for _ in range(self.num_loc_convs)])
        self.seg_convs = nn.ModuleList(
            [_ConvGNReLU(self.in_channels, self.out_channels, 1, norm_cfg=self.norm_cfg) for _ in range(self.num_seg_convs)])

    def init_weights(self):
        """knet/det/kernel_head.py:170-202."""
        if self.localization_fpn is not None and hasattr(self.localization_fpn, 'init_weights'):
            self.localization_fpn.init_weights()
        if self.feat_downsample_stride > 1 and self.conv_normal_init:
            for conv in [self.loc_convs, self.seg_convs]:
                for m in conv.modules():
                    if isinstance(m, nn.Conv2d):
                        nn.init.normal_(m.weight, 0, 0.01)
        if self.semantic_fpn:
            nn.init.normal_(self.conv_seg.weight, 0, 0.01)
            use_sigmoid = bool((self.loss_cfgs.get('loss_seg') or {}).get('use_sigmoid', False))
            nn.init.constant_(self.conv_seg.bias, float(-torch.log(torch.tensor((1 - 0.01) / 0.01))) if use_sigmoid else 0.0)
        if self.xavier_init_kernel:
            nn.init.xavier_uniform_(self.init_kernels.weight)
        else:
            nn.init.normal_(self.init_kernels.weight, 0, self.kernel_init_std)

    # ------------------------------------------------------------------------------------------------------------------
    def _upstream_feats(self, img):
        """localization FPN + loc/seg convs (+ downsample): the backbone-side part, plain torch modules
        (knet/det/kernel_head.py:207-230)."""
        if self.localization_fpn is None:
            raise NotImplementedError('no localization_fpn was given: call decode_init_proposals_from_feats(loc, sem)')
        feats = self.localization_fpn(img)
        loc = feats[0] if isinstance(feats, (list, tuple)) else feats
        for conv in self.loc_convs:
            loc = conv(loc)
        if self.feat_downsample_stride > 1 and self.feat_refine:
            loc = self.ins_downsample(loc)
        sem = None
        if self.semantic_fpn:
            sem = feats[1] if isinstance(feats, (list, tuple)) else feats
            for conv in self.seg_convs:
                sem = conv(sem)
            if self.feat_downsample_stride > 1 and self.feat_refine:
                sem = self.seg_downsample(sem)
        return loc, sem

    def decode_init_proposals_from_feats(self, loc_feats, semantic_feats=None):
        """The hot part of `_decode_init_proposals` (knet/det/kernel_head.py:221-263) in one C-ABI call.
        Returns the reference's 5-tuple `(proposal_feats [B,N,C,1,1], x_feats, mask_preds, cls_scores=None, seg_preds)`."""
        if torch.is_grad_enabled() and (loc_feats.requires_grad or (semantic_feats is not None and semantic_feats.requires_grad)
                                        or any(p.requires_grad for p in (self.init_kernels.weight,))):
            return self._decode_autograd(loc_feats, semantic_feats)
        cat = self.cat_stuff_mask and not self.training
        prop, x_feats, mask_preds, seg_preds = ops.kernel_init(
            loc_feats, semantic_feats if self.semantic_fpn else None, self.init_kernels.weight,
            self.conv_seg.weight if self.semantic_fpn else None, self.conv_seg.bias if self.semantic_fpn else None,
            num_thing_classes=self.num_thing_classes, cat_stuff_mask=cat, proposal_feats_with_obj=self.proposal_feats_with_obj,
            use_binary=self.use_binary)
        return prop.reshape(*prop.shape, 1, 1), x_feats, mask_preds, None, seg_preds

    def _decode_init_proposals(self, img, img_metas):
        loc, sem = self._upstream_feats(img)
        return self.decode_init_proposals_from_feats(loc, sem)

    def simple_test_rpn(self, img, img_metas):
        """knet/det/kernel_head.py:506-508."""
        with torch.no_grad():
            return self._decode_init_proposals(img, img_metas)

    def forward_dummy(self, img, img_metas):
        """knet/det/kernel_head.py:510-515."""
        return self._decode_init_proposals(img, img_metas)

    # ------------------------------------------------------------------------------------------------------------------ training
    def _decode_autograd(self, loc_feats, semantic_feats):
        """Differentiable `_decode_init_proposals` after the loc / seg convs (knet/det/kernel_head.py:221-263): the two 1x1 convs are
        the HIP decode (frame-shared kernels; the expand's backward sums their gradient over the frames), the object features the
        HIP gather — both through their autograd wrappers (video-k-net_amd/autograd.py).  The stuff concatenation is the eval
        branch (:255-263); in training `forward_train` appends the stuff kernels itself (:326-334)."""
        B, C = loc_feats.shape[:2]
        Np = self.num_proposals
        w = self.init_kernels.weight.reshape(Np, C)
        mask_preds = vag.mask_decode(loc_feats, w[None].expand(B, Np, C))                                  # :222
        seg_preds = None
        if self.semantic_fpn:
            sw = self.conv_seg.weight.reshape(self.num_classes, C)
            seg_preds = vag.mask_decode(semantic_feats, sw[None].expand(B, self.num_classes, C),
                                        self.conv_seg.bias[None].expand(B, self.num_classes))               # :231-234
        proposal_feats = self.init_kernels.weight[None].expand(B, Np, C, 1, 1)                             # :234-236
        x_feats = semantic_feats + loc_feats if semantic_feats is not None else loc_feats                  # :238-241
        if self.proposal_feats_with_obj:
            if self.use_binary:
                obj, _ = vag.mask_gather(x_feats, mask_preds.detach(), 0.5)                                # :243-250 (bool mask: no grad)
            else:
                obj = vag.mask_gather_soft(x_feats, mask_preds, 0.5)                                       # :246-249: gradients reach the logits
            proposal_feats = proposal_feats + obj.view(B, Np, C, 1, 1)                                     # :252-254
        if self.cat_stuff_mask and not self.training:                                                       # :255-263
            mask_preds = torch.cat([mask_preds, seg_preds[:, self.num_thing_classes:]], dim=1)
            stuff = self.conv_seg.weight[self.num_thing_classes:].clone()
            proposal_feats = torch.cat([proposal_feats, stuff[None].expand(B, *stuff.shape)], dim=1)
        return proposal_feats, x_feats, mask_preds, None, seg_preds

    def forward_train(self, img, img_metas, gt_masks, gt_labels, gt_sem_seg=None, gt_sem_cls=None):
        """-> (losses, proposal_feats, x_feats, mask_preds, cls_scores)                           knet/det/kernel_head.py:267-336"""
        if not self.train_cfg:
            raise RuntimeError('forward_train needs train_cfg (assigner / pos_weight)')
        num_imgs = len(img_metas)
        proposal_feats, x_feats, mask_preds, cls_scores, seg_preds = self._decode_init_proposals(img, img_metas)
        s_ = self.feat_downsample_stride
        if s_ > 1:
            scaled_mask_preds = F.interpolate(mask_preds, scale_factor=s_, mode='bilinear', align_corners=False)
            scaled_seg_preds = (F.interpolate(seg_preds, scale_factor=s_, mode='bilinear', align_corners=False)
                                if seg_preds is not None else None)
        else:
            scaled_mask_preds, scaled_seg_preds = mask_preds, seg_preds
        if self.hard_target:
            gt_masks = [g.bool().float() for g in gt_masks]
        sampling_results = []
        for i in range(num_imgs):
            assign_result = self.assigner.assign(scaled_mask_preds[i].detach(), None, gt_masks[i], gt_labels[i], img_metas[i])
            sampling_results.append(self.sampler.sample(assign_result, scaled_mask_preds[i], gt_masks[i]))
        mask_targets = self._targets(sampling_results, self.train_cfg, True, gt_sem_seg, gt_sem_cls)
        losses = self.loss(scaled_mask_preds, cls_scores, scaled_seg_preds, proposal_feats, *mask_targets)
        if hasattr(self.assigner, 'check_status'):
            self.assigner.check_status(wait=False)     # device LSAP status words -> the asynchronous flag queue (no stall)
        if self.cat_stuff_mask and self.training:
            mask_preds = torch.cat([mask_preds, seg_preds[:, self.num_thing_classes:]], dim=1)
            stuff_kernels = self.conv_seg.weight[self.num_thing_classes:].clone()
            proposal_feats = torch.cat([proposal_feats, stuff_kernels[None].expand(num_imgs, *stuff_kernels.size())], dim=1)
        return losses, proposal_feats, x_feats, mask_preds, cls_scores

    def loss(self, mask_pred, cls_scores, seg_preds, proposal_feats, labels, label_weights, mask_targets, mask_weights, seg_targets,
             reduction_override=None, **kwargs):
        """`loss_rpn_mask / _dice / _rank / _seg` (+ `_cls`, `rpn_pos_acc` when the head scores classes)      :338-425"""
        losses = dict()
        bg = self.num_classes
        pos = (labels >= 0) & (labels < bg)
        num_preds = mask_pred.shape[0] * mask_pred.shape[1]
        if cls_scores is not None:
            avg_factor = reduce_mean(pos.sum().float())
            losses['loss_rpn_cls'] = self.loss_cls(cls_scores.view(num_preds, -1), labels, label_weights, avg_factor=avg_factor,
                                                   reduction_override=reduction_override)
            losses['rpn_pos_acc'] = accuracy(cls_scores.view(num_preds, -1)[pos], labels[pos])
        H, W = mask_pred.shape[-2:]
        if pos.any():
            pos_mask_pred = mask_pred.reshape(num_preds, H, W)[pos]
            pos_mask_targets = mask_targets[pos]
            losses['loss_rpn_mask'] = self.loss_mask(pos_mask_pred, pos_mask_targets)
            losses['loss_rpn_dice'] = self.loss_dice(pos_mask_pred, pos_mask_targets)
            if self.loss_rank is not None:
                # every pixel gets the LARGEST positive kernel index whose target covers it (the reference paints the targets in
                # ascending kernel order, :375-386) — a masked max over the kernel axis instead of a per-instance host loop
                B = mask_pred.size(0)
                n = mask_targets.shape[0] // B
                covered = mask_targets.view(B, n, H, W).bool() & pos.view(B, n, 1, 1)
                idx = torch.arange(n, device=mask_targets.device, dtype=torch.int16).view(1, n, 1, 1)
                top = torch.where(covered, idx, idx.new_full((), -1)).amax(dim=1).long()
                rank_target = torch.where(top >= 0, top, top.new_full((), self.ignore_label))
                losses['loss_rpn_rank'] = self.loss_rank(mask_pred, rank_target, ignore_index=self.ignore_label)
        else:
            losses['loss_rpn_mask'] = mask_pred.sum() * 0
            losses['loss_rpn_dice'] = mask_pred.sum() * 0
            if self.loss_rank is not None:
                losses['loss_rank'] = mask_pred.sum() * 0                                     # (the reference's key, :393)
        if seg_preds is not None:
            ch = seg_preds.shape[1]
            flat = seg_preds.view(-1, ch, H * W).permute(0, 2, 1).reshape(-1, ch)
            tgt = seg_targets.view(-1)
            if self.loss_seg.use_sigmoid:                                                       # focal (:397-410)
                dense_pos = ((tgt >= 0) & (tgt < bg)).sum().float().clamp(min=1.0)
                losses['loss_rpn_seg'] = self.loss_seg(flat, tgt, avg_factor=dense_pos)
            else:                                                                               # ce (:412-418)
                losses['loss_rpn_seg'] = self.loss_seg(flat, tgt, ignore_index=self.num_classes)
        return losses

    def _image_targets(self, n, shape, dtype, dev, pos_inds, pos_gt_mask, pos_gt_labels, gt_sem_seg, gt_sem_cls, cfg):
        """Targets of one image (reference :427-466) from what defines them — the matched rows with their ground truth — without the
        matched / unmatched mask copies the reference passes around:
          labels [n] (background = num_classes), label_weights [n] (matched: pos_weight if > 0, everything else 1),
          mask_targets [n, H, W] and mask_weights (1 on the matched rows; an expanded view of a [n] vector),
          seg_targets [H, W]: the dense semantic target.  The reference PAINTS it — the stuff masks in order, then every matched
          instance's label over its mask in sample order — so a pixel ends up with the label of the LAST layer covering it; here
          that is one masked arg-max over the stacked layers (no per-instance boolean-mask writes, each of which synchronises)."""
        H, W = shape
        k = int(pos_inds.shape[0])
        labels = torch.full((n,), self.num_classes, dtype=torch.long, device=dev)
        label_weights = torch.ones((n,), dtype=dtype, device=dev)       # unmatched rows weigh 1, matched ones pos_weight
        mask_targets = torch.zeros((n, H, W), dtype=dtype, device=dev)
        row_w = torch.zeros((n,), dtype=dtype, device=dev)
        layers, layer_labels = [], []
        if gt_sem_cls is not None and gt_sem_seg is not None and len(gt_sem_cls) > 0:
            layers.append(gt_sem_seg.to(dev).bool())
            layer_labels.append(gt_sem_cls.to(dev).long())
        if k > 0:
            pw = self._cfg(cfg, 'pos_weight')
            labels[pos_inds] = pos_gt_labels
            if pw > 0 and pw != 1:
                label_weights.index_fill_(0, pos_inds, float(pw))
            mask_targets[pos_inds] = pos_gt_mask.to(dtype)
            row_w.index_fill_(0, pos_inds, 1.0)
            layers.append(pos_gt_mask.bool())
            layer_labels.append(pos_gt_labels.long())
        if layers:
            stack, lab = torch.cat(layers), torch.cat(layer_labels)
            order = torch.arange(stack.shape[0], device=dev, dtype=torch.int32).view(-1, 1, 1)
            top = torch.where(stack, order, order.new_full((), -1)).amax(dim=0).long()
            seg_targets = torch.where(top >= 0, lab[top.clamp(min=0)], lab.new_full((), self.num_classes))
        else:
            seg_targets = torch.full((H, W), self.num_classes, dtype=torch.long, device=dev)
        return labels, label_weights, mask_targets, row_w.view(-1, 1, 1).expand(-1, H, W), seg_targets

    def _get_target_single(self, pos_inds, neg_inds, pos_mask, neg_mask, pos_gt_mask, pos_gt_labels, gt_sem_seg, gt_sem_cls, cfg):
        """The reference's per-image entry point (:427-466), kept for callers that hold its argument list."""
        return self._image_targets(pos_mask.size(0) + neg_mask.size(0), tuple(pos_mask.shape[-2:]), pos_mask.dtype, pos_mask.device,
                                   pos_inds, pos_gt_mask, pos_gt_labels, gt_sem_seg, gt_sem_cls, cfg)

    def get_targets(self, sampling_results, gt_mask, rpn_train_cfg, concat=True, gt_sem_seg=None, gt_sem_cls=None):
        """:468-504 (`gt_mask` is not read there either: the sampling results carry the matched ground truth)"""
        return self._targets(sampling_results, rpn_train_cfg, concat, gt_sem_seg, gt_sem_cls)

    def _targets(self, sampling_results, rpn_train_cfg, concat, gt_sem_seg, gt_sem_cls):
        n = len(sampling_results)
        if gt_sem_seg is None:
            gt_sem_seg, gt_sem_cls = [None] * n, [None] * n      # (the reference hard-codes 2, :480-481)
        out = [self._image_targets(r.num_pos + r.num_neg, r.mask_shape[-2:], r.mask_dtype, r.device, r.pos_inds, r.pos_gt_masks, r.pos_gt_labels, gt_sem_seg[i], gt_sem_cls[i], rpn_train_cfg)
               for i, r in enumerate(sampling_results)]
        labels, label_weights, mask_targets, mask_weights, seg_targets = (list(t) for t in zip(*out))
        if concat:
            labels, label_weights = torch.cat(labels, 0), torch.cat(label_weights, 0)
            mask_targets, mask_weights = torch.cat(mask_targets, 0), torch.cat(mask_weights, 0)
            seg_targets = torch.stack(seg_targets, 0)
        return labels, label_weights, mask_targets, mask_weights, seg_targets


@register_head
class ConvKernelHeadVideo(ConvKernelHead):
    """knet_vis/tracker/kernel_head.py:12-512 — the kernel-initialisation head of the VIS models (`rpn_head` of
    configs/video_knet_vis/video_knet_vis/knet_track_*_youtubevis.py).  It is `ConvKernelHead` over the `B * T` frames of a batch of
    `B` clips: `img` / the feature maps arrive flattened to `[B*T, ...]`, every frame gets the same initial kernels, and what
    differs from the per-image head is bookkeeping only — per-frame metas come as `ref_img_metas[clip][frame]`, ground-truth masks
    as `gt_masks[clip][frame]` and labels as `gt_labels[clip]` rows `(frame index, label)` (:303-317).  So the frames are
    flattened here and everything — HIP kernel-init pass, assignment, targets, losses — is the base class's."""

    def _init_layers(self):
        super()._init_layers()
        if self.semantic_fpn and not self.loss_seg.use_sigmoid:
            # the reference gives the semantic branch a background channel then (:131-136); no shipped config does
            raise NotImplementedError('ConvKernelHeadVideo with a soft-max semantic loss (num_classes + 1 channels) is not built')

    @staticmethod
    def _frame_metas(img_metas, ref_img_metas):
        if ref_img_metas is None:
            return img_metas
        return [m for clip in ref_img_metas for m in clip]

    def _decode_init_proposals(self, img, img_metas, ref_img_metas=None):
        return super()._decode_init_proposals(img, self._frame_metas(img_metas, ref_img_metas))

    def simple_test_rpn(self, img, img_metas, ref_img_metas=None):
        with torch.no_grad():
            return self._decode_init_proposals(img, img_metas, ref_img_metas)

    def get_targets(self, sampling_results, rpn_train_cfg, concat=True, gt_sem_seg=None, gt_sem_cls=None):
        """(:466-501: the VIS head's signature has no `gt_mask`.)"""
        return self._targets(sampling_results, rpn_train_cfg, concat, gt_sem_seg, gt_sem_cls)

    def forward_dummy(self, img, img_metas, ref_img_metas=None):
        return self._decode_init_proposals(img, img_metas, ref_img_metas)

    def forward_train(self, img, img_metas, ref_img_metas, gt_masks, gt_labels, gt_instance_ids=None, gt_sem_seg=None,
                      gt_sem_cls=None):
        """-> (losses, proposal_feats, x_feats, mask_preds, cls_scores) over the B*T frames (:267-334)."""
        metas = self._frame_metas(img_metas, ref_img_metas)
        flat_masks, flat_labels = [], []
        for i, clip in enumerate(ref_img_metas):
            rows = gt_labels[i]
            for j in range(len(clip)):
                flat_masks.append(gt_masks[i][j])
                flat_labels.append(rows[:, 1][rows[:, 0] == j])
        return super().forward_train(img, metas, flat_masks, flat_labels, gt_sem_seg=gt_sem_seg, gt_sem_cls=gt_sem_cls)


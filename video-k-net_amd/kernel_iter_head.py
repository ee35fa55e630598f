"""`KernelIterHead` / `VideoKernelIterHead` — drop-ins for the reference's S-stage iteration
(knet/det/kernel_iter_head.py:11-311, knet/video/kernel_iter_head.py:10-564): same `HEADS` registration, ctor kwargs,
`mask_head.{s}.*` module tree and call signatures.  The stage loop itself is ONE C-ABI call (`vkn_head_forward_f32`) when
every stage is one of our heads on a GPU tensor; `_mask_forward` exposes the per-stage path exactly like the reference.

The post-head panoptic path (`simple_test` with do_panoptic + merge_joint, `get_panoptic`) runs the fused
`vkn_panoptic_joint_f32` pipeline straight from the low-res mask logits (no [K, H, W] full-resolution tensor is materialised).
The thing-first merge (`merge_joint=False`: `merge_stuff_thing`, on the device through `vkn_panoptic_thing_first_u8`), the
instance-only `get_seg_masks` path (`do_panoptic=False`) and training (`forward_train`: GPU assignment costs + C++ LSAP, torch
losses, autograd through the HIP gather / decode kernels) are provided as well.
"""
import numpy as np
import torch
import torch.nn as nn

from . import ops
from .kernel_update_head import KernelUpdateHead, VideoKernelUpdateHead
from .registry import BaseRoIHead, build_assigner, build_head, build_sampler, register_head


@register_head
class KernelIterHead(BaseRoIHead):

    def __init__(self, num_stages=6, recursive=False, assign_stages=5, stage_loss_weights=(1, 1, 1, 1, 1, 1),
                 proposal_feature_channel=256, merge_cls_scores=False, do_panoptic=False, post_assign=False,
                 hard_target=False, merge_joint=False, num_proposals=100, num_thing_classes=80, num_stuff_classes=53,
                 mask_assign_stride=4, ignore_label=255, thing_label_in_seg=0,
                 mask_head=dict(type='KernelUpdateHead', num_classes=80, num_fcs=2, num_heads=8, num_cls_fcs=1,
                                num_reg_fcs=3, feedforward_channels=2048, hidden_channels=256, dropout=0.0,
                                roi_feat_size=7, ffn_act_cfg=dict(type='ReLU', inplace=True)),
                 mask_out_stride=4, train_cfg=None, test_cfg=None, **kwargs):
        assert mask_head is not None
        assert len(stage_loss_weights) == num_stages
        self.num_stages = num_stages
        self.stage_loss_weights = stage_loss_weights
        self.proposal_feature_channel = proposal_feature_channel
        self.merge_cls_scores = merge_cls_scores
        self.recursive = recursive
        self.post_assign = post_assign
        self.mask_out_stride = mask_out_stride
        self.hard_target = hard_target
        self.assign_stages = assign_stages
        self.do_panoptic = do_panoptic
        self.merge_joint = merge_joint
        self.num_thing_classes = num_thing_classes
        self.num_stuff_classes = num_stuff_classes
        self.num_classes = self.num_thing_classes + self.num_stuff_classes
        self.mask_assign_stride = mask_assign_stride
        self.thing_label_in_seg = thing_label_in_seg
        self.num_proposals = num_proposals
        self.ignore_label = ignore_label
        self._init_extra(kwargs)
        super().__init__(mask_head=mask_head, train_cfg=train_cfg, test_cfg=test_cfg, **kwargs)

    def _init_extra(self, kwargs):
        pass

    # ---- BaseRoIHead contract (reference :76-116)
    def init_bbox_head(self, mask_roi_extractor, mask_head):
        pass

    def init_assigner_sampler(self):
        """One assigner + sampler per stage from `train_cfg` (a list of per-stage dicts; reference :85-95)."""
        self.mask_assigner = []
        self.mask_sampler = []
        if self.train_cfg is not None:
            for idx, rcnn_train_cfg in enumerate(self.train_cfg):
                self.mask_assigner.append(build_assigner(self._cfg(rcnn_train_cfg, 'assigner')))
                self.current_stage = idx
                self.mask_sampler.append(build_sampler(self._cfg(rcnn_train_cfg, 'sampler'), context=self))

    def init_weights(self):
        for i in range(self.num_stages):
            self.mask_head[i].init_weights()

    def enable_chain_graphs(self, on=True):
        """Training: run every stage's [B*N, C] chain (forward and backward) as captured hipGraphs — see
        `KernelUpdateHead.enable_chain_graphs`.  The gather / decode kernels, the assignment and the losses stay as they are."""
        for h in self.mask_head:
            h.enable_chain_graphs(on)
        return self

    def _chain_graphs_new_step(self):
        for h in self.mask_head:
            if hasattr(h, 'chain_graphs_new_step'):
                h.chain_graphs_new_step()

    def init_mask_head(self, mask_roi_extractor, mask_head):
        self.mask_head = nn.ModuleList()
        if not isinstance(mask_head, list):
            mask_head = [mask_head for _ in range(self.num_stages)]
        assert len(mask_head) == self.num_stages
        for head in mask_head:
            self.mask_head.append(build_head(head))
        if self.recursive:
            for i in range(self.num_stages):
                self.mask_head[i] = self.mask_head[0]

    # ---- per-stage path (reference :118-137)
    def _mask_forward(self, stage, x, object_feats, mask_preds, img_metas):
        mask_head = self.mask_head[stage]
        cls_score, mask_preds, object_feats = mask_head(x, object_feats, mask_preds, img_metas=img_metas)
        if mask_head.mask_upsample_stride > 1 and (stage == self.num_stages - 1 or self.training):
            scaled_mask_preds = self._upsample(mask_preds, mask_head.mask_upsample_stride)
        else:
            scaled_mask_preds = mask_preds
        return dict(cls_score=cls_score, mask_preds=mask_preds, scaled_mask_preds=scaled_mask_preds,
                    object_feats=object_feats)

    _lowres_tail_step = False     # set by `_train_stages` for the duration of a step that runs the low-res loss tail
    _defer_scaled = False         # ... and for its stages whose up-scaled logits nobody outside the loop sees (every stage but the last)

    class DeferredScaled:
        """The x`stride` up-scaling of a stage's low-res mask logits that has NOT been computed: inside the training loop the
        assignment costs and the loss tail work from the low-res logits (vkn_assign_costs_lowres_batch_f32,
        vkn_mask_losses_fwd_lowres_f32 / _bwd_lowres_f32), so the [B, Ns, S h, S w] tensor — 981 MB per stage at the shipped x4 —
        exists only if something asks for it: `materialize()` (values + the adjoint as backward, as `_upsample` returns)."""

        def __init__(self, lowres, stride, make):
            self._vkn_lowres, self.stride, self._make, self._t = lowres, stride, make, None
            b, n, h, w = lowres.shape
            self.shape = torch.Size((b, n, h * stride, w * stride))
            self.dtype, self.device, self.is_cuda = lowres.dtype, lowres.device, lowres.is_cuda

        def dim(self):
            return 4

        def detach(self):
            return self

        def materialize(self):
            if self._t is None:
                self._t = self._make(self._vkn_lowres, self.stride)
            return self._t

        def __getitem__(self, i):
            return self.materialize()[i]

    def _upsample(self, mask_preds, stride):
        """`F.interpolate(mask_preds, scale_factor=stride, bilinear, align_corners=False)` (reference :122-130): the HIP kernel; under
        autograd the same kernel with its adjoint as backward."""
        if self._lowres_tail_step and self._defer_scaled and mask_preds.requires_grad and torch.is_grad_enabled():
            return self.DeferredScaled(mask_preds, stride, self._upsample_now)
        return self._upsample_now(mask_preds, stride)

    def _upsample_now(self, mask_preds, stride):
        if self._lowres_tail_step and mask_preds.requires_grad and torch.is_grad_enabled():
            # the fused loss tail differentiates w.r.t. the LOW-RES logits itself (train_tail.py, vkn_mask_losses_bwd_lowres_f32): the
            # up-scaled values carry no graph of their own — but whoever back-propagates through the RETURNED tensor (nobody in the
            # reference: the detector takes boxes from it) still reaches the logits: `LazyUpsampleFn` = these values + the adjoint
            from . import autograd as vag
            return vag.lazy_upsample(mask_preds, ops.upsample_bilinear(mask_preds.detach(), stride), stride)
        if mask_preds.requires_grad and torch.is_grad_enabled():
            from . import autograd as vag
            if mask_preds.is_cuda and mask_preds.dtype == torch.float32 and mask_preds.dim() == 4:
                out = vag.upsample_bilinear(mask_preds, stride)       # HIP forward + its adjoint (csrc/vkn_loss.hip)
            else:
                out = torch.nn.functional.interpolate(mask_preds, scale_factor=stride, mode='bilinear', align_corners=False)
        else:
            out = ops.upsample_bilinear(mask_preds, stride)
        out._vkn_lowres = mask_preds     # (witness for `_train_stages`: these values are the bilinear x`stride` up-scaling of exactly this tensor)
        return out

    def check_status(self, device=None):
        """Raise `VknError` (VKN_E_RANGE) when a call since the last check fed the kernels features outside the f16-split envelope
        (|x| >= 65504 or non-finite): `ops.workspace_status`.  One stream synchronisation."""
        ops.workspace_status(device)

    # ---- fused S-stage loop
    def _fused_ok(self, x):
        # the `simple_test*` entry points are INFERENCE APIs: in eval() mode they take the fused C-ABI path and their outputs are
        # detached from autograd whatever the grad mode (mmdet calls them under no_grad); gradients flow through `forward_train`
        # and through the stage modules' `forward` (KernelUpdateHead._needs_grad)
        # (a softmax classification head — `loss_cls.use_sigmoid=False`, reference :309-310, no shipped config — runs stage by stage:
        #  the fused call applies the sigmoid in its last epilogue)
        return (x.is_cuda and not self.training
                and all(isinstance(h, KernelUpdateHead) for h in self.mask_head)
                and self.mask_head[-1].loss_cls.use_sigmoid
                and len({(h.in_channels, h.num_heads, h.feedforward_channels, h.fc_cls.out_features, h.num_cls_fcs,
                          h.num_mask_fcs, h.hard_mask_thr, h.with_ffn, h.feat_transform is None) for h in self.mask_head}) == 1)

    def _head_forward(self, x, proposal_feats, mask_preds, previous_obj_feats=None, want_track=False, flags=0,
                      want_scaled=True, clip_first_prev=None):
        h0, hl = self.mask_head[0], self.mask_head[-1]
        for h in self.mask_head:
            h._check_inputs(x, proposal_feats, mask_preds, None)
        mask_preds = h0._gather_masks(x, mask_preds)       # (reference kernel_update_head.py:182-188: only the incoming masks can differ)
        B, N = proposal_feats.shape[:2]
        C, K = h0.in_channels, h0.conv_kernel_size
        H, W = x.shape[-2:]
        dims = h0.make_dims(B, N, H, W)
        packs = [h.stage_pack(x.device) for h in self.mask_head]
        prev = previous_obj_feats.reshape(B, N, C) if previous_obj_feats is not None else None
        # previous_link / previous_type="update" blocks of the LAST stage (knet/video/kernel_iter_head.py:544-546)
        link_pre, link_track, track_src = hl.link_packs(x.device) if (prev is not None or clip_first_prev is not None) else (None, None, 0)
        obj, cls, masks, scaled, track = ops.head_forward(dims, packs, x, proposal_feats.reshape(B, N, C), mask_preds, prev,
                                                          hl.mask_upsample_stride, want_track=want_track, want_scaled=want_scaled,
                                                          flags=flags | getattr(h0, 'vkn_flags', 0), clip_first_prev=clip_first_prev, link_pre=link_pre,
                                                          link_track=link_track, track_src=track_src)
        if not hl.loss_cls.use_sigmoid:
            # (the reference entry points never get here: `_fused_ok` sends a softmax head through the stage-by-stage path; the
            #  clip-batched extensions have no such path)
            raise NotImplementedError('the fused head call applies sigmoid to the class logits; a softmax head (reference :309-310) '
                                      'runs through simple_test_mask_preds*')
        obj = obj.reshape(B, N, C, K, K)
        if track is not None:
            track = track.reshape(B, N, C, K, K)
        return obj, cls, masks, scaled, track

    def simple_test_mask_preds(self, x, proposal_feats, mask_preds, cls_score, img_metas, imgs_whwh=None, rescale=False):
        """-> (object_feats, cls_score.sigmoid(), mask_preds, scaled_mask_preds)            reference :285-311"""
        if self._fused_ok(x):
            return self._head_forward(x, proposal_feats, mask_preds)[:4]
        object_feats = proposal_feats
        for stage in range(self.num_stages):
            r = self._mask_forward(stage, x, object_feats, mask_preds, img_metas)
            object_feats, cls_score = r['object_feats'], r['cls_score']
            mask_preds, scaled_mask_preds = r['mask_preds'], r['scaled_mask_preds']
        if self.mask_head[-1].loss_cls.use_sigmoid:
            cls_score = cls_score.sigmoid()
        else:
            cls_score = cls_score.softmax(-1)[..., :-1]
        return object_feats, cls_score, mask_preds, scaled_mask_preds

    def forward_dummy(self, x, proposal_boxes, proposal_feats, img_metas):
        """FLOPs-counting harness of the reference (:316-330)."""
        num_imgs = len(img_metas)
        C = x.shape[1]
        mask_preds = ops.mask_decode(x, proposal_feats.reshape(num_imgs, -1, C))
        object_feats, out = proposal_feats, []
        for stage in range(self.num_stages):
            r = self._mask_forward(stage, x, object_feats, mask_preds, img_metas)
            out.append(r)
        return out

    x_hub = not __import__('os').environ.get('VKN_NO_XHUB')     # (A/B switch of autograd.x_hub)
    fused_tail = True     # False: per-image sampler -> get_targets -> loss, op by op (A/B; taken anyway whenever train_tail.TailStep declines)
    lowres_tail = True    # the fused tail's backward pass straight into the low-res logits (False: x`up` gradient + upsample adjoint, A/B)

    def _train_stages(self, x, proposal_feats, mask_preds, cls_score, img_metas, gt_masks, gt_labels, imgs_whwh=None,
                      gt_sem_seg=None, gt_sem_cls=None, stage_kwargs=None):
        """The per-stage assign -> sample -> targets -> loss loop shared by `forward_train` (reference :139-231) and the video
        head's `forward_train[_with_previous]` (knet/video/kernel_iter_head.py:150-376).  Assignment runs on the GPU cost kernels
        + the C++ LSAP (`MaskHungarianAssigner`), losses in torch autograd, the stage forward through the HIP kernels' autograd
        wrappers.  `stage_kwargs(stage)` gives extra keyword arguments of `_mask_forward` (the last stage's previous-frame link)."""
        if not self.mask_assigner:
            raise RuntimeError('forward_train needs train_cfg (one dict per stage with assigner / sampler / pos_weight)')
        num_imgs = len(img_metas)
        self._chain_graphs_new_step()
        up = self.mask_head[0].mask_upsample_stride
        # what stage s is assigned on: the predictions it RECEIVES (reference :150-156, :225-226) — or, with `post_assign`, its own
        # — as the low-res logits whose x`up` up-scaling the reference assigns on: the assigner's cost kernel interpolates them itself
        # (vkn_assign_costs_lowres_batch_f32); the up-scaled tensor (`assign_masks`) is formed only if the assigner declines
        assign_low = mask_preds.detach() if up > 1 else None
        assign_masks = None if up > 1 else mask_preds.detach()
        assign_cls = cls_score.detach() if cls_score is not None else None
        if self.hard_target:
            gt_masks = [g.bool().float() for g in gt_masks]
        object_feats = proposal_feats
        all_stage_loss, assign_results, mask_results = {}, None, None
        # the fused loss tail (train_tail.py): the batch's ground truth as one bank, targets and losses per stage on the library's
        # kernels; None whenever a precondition fails — then the op-by-op path below runs (same values)
        if self.fused_tail and self.x_hub and torch.is_grad_enabled() and torch.is_tensor(x) and x.is_cuda and x.requires_grad:
            from . import autograd as vag
            x = vag.x_hub(x)        # the six gradient contributions of x (a gather + a decode per stage) are summed in one pass
        from .train_tail import TailStep
        tail = TailStep.begin(self, x.device, gt_masks, gt_labels, gt_sem_seg, gt_sem_cls)
        if tail is not None:
            gt_masks = tail.gt_views
        self._last_tail_fused = tail is not None     # (tests / bench: did the step's EVERY stage run the fused tail?)
        # the low-res form of the tail's backward (no x`up` gradient tensor, no upsample adjoint): strides 2 / 4, plain fp32 CUDA logits
        lowres = tail is not None and self.lowres_tail and up in (2, 4) and mask_preds.is_cuda and mask_preds.dtype == torch.float32
        self._lowres_tail_step = bool(lowres)
        if self.mask_assigner and hasattr(self.mask_assigner[0], 'validate_labels'):
            # the labels do not change between stages: one range check for the whole step (on the device, reported asynchronously)
            if tail is not None:
                self.mask_assigner[0].validate_labels(gt_labels, self.num_thing_classes, status=tail.status)
            else:
                self.mask_assigner[0].validate_labels(gt_labels, self.num_thing_classes)
        for stage in range(self.num_stages):
            extra = stage_kwargs(stage) if stage_kwargs is not None else {}
            # the up-scaled logits of every stage but the last stay uncomputed unless something below asks for them
            self._defer_scaled = bool(lowres) and stage < self.num_stages - 1 and getattr(tail, 'lowres_forward', False)
            mask_results = self._mask_forward(stage, x, object_feats, mask_preds, img_metas, **extra)
            self._defer_scaled = False
            mask_preds, scaled_mask_preds = mask_results['mask_preds'], mask_results['scaled_mask_preds']
            cls_score, object_feats = mask_results['cls_score'], mask_results['object_feats']
            stage_low = mask_preds.detach() if up > 1 and getattr(scaled_mask_preds, '_vkn_lowres', None) is mask_preds else None
            if self.post_assign:
                assign_masks, assign_cls, assign_low = (None if stage_low is not None else scaled_mask_preds.detach()), cls_score.detach(), stage_low
            if stage < self.assign_stages:       # later stages keep the last assignment (:196)
                assign_results = self._assign_batch(stage, assign_masks, assign_cls, gt_masks, gt_labels, img_metas,
                                                    lowres=(assign_low, up) if assign_low is not None else None)
            head = self.mask_head[stage]
            stage_losses = None
            if tail is not None and scaled_mask_preds.shape[1] == self.num_proposals + (head.num_stuff_classes if tail.with_sem else 0) \
                    and tail.stage_ok(head, assign_results, cls_score, scaled_mask_preds):
                if lowres and getattr(scaled_mask_preds, '_vkn_lowres', None) is mask_preds:
                    stage_losses = tail.stage_losses(head, self.train_cfg[stage], assign_results, cls_score, scaled_mask_preds, lowres=mask_preds,
                                                     stride=up)
                else:
                    stage_losses = tail.stage_losses(head, self.train_cfg[stage], assign_results, cls_score, scaled_mask_preds)
            if stage_losses is None:
                self._last_tail_fused = False
                if hasattr(scaled_mask_preds, 'materialize'):
                    scaled_mask_preds = mask_results['scaled_mask_preds'] = scaled_mask_preds.materialize()
                sampler = self.mask_sampler[stage]
                sampling_results = [sampler.sample(assign_results[i], scaled_mask_preds[i], gt_masks[i]) for i in range(num_imgs)]
                mask_targets = head.get_targets(sampling_results, gt_masks, gt_labels, self.train_cfg[stage], True,
                                                gt_sem_seg=gt_sem_seg, gt_sem_cls=gt_sem_cls)
                stage_losses = head.loss(object_feats, cls_score, scaled_mask_preds, *mask_targets, imgs_whwh=imgs_whwh)
            w = self.stage_loss_weights[stage]
            for key, value in stage_losses.items():
                all_stage_loss[f's{stage}_{key}'] = value if w == 1 else value * w      # (x * 1 == x: no launch, no autograd node)
            if not self.post_assign:
                assign_masks, assign_cls, assign_low = (None if stage_low is not None else scaled_mask_preds.detach()), cls_score.detach(), stage_low
        self._lowres_tail_step = False
        if tail is not None:
            tail.finish()
        if self.mask_assigner and hasattr(self.mask_assigner[0], 'check_status'):
            # device assignments report invalid cost matrices through status words: ONE read per step for all stages
            self.mask_assigner[0].check_status(*self.mask_assigner[1:], wait=False)   # (reported at a later poll: no stall)
        return all_stage_loss, mask_results

    def _assign_batch(self, stage, masks, cls, gt_masks, gt_labels, img_metas, lowres=None):
        """One-to-one assignment of a stage for every image of the batch: proposals only (the stuff kernels have fixed targets), thing
        logits only.  One LSAP launch for the batch when the assigner offers it (`MaskHungarianAssigner.assign_batch`).
        `lowres = ([B, Ns, h, w] logits, stride)`: what the reference assigns on is exactly their bilinear x`stride` up-scaling;
        `masks` (that up-scaling) may then be None — it is formed here only if the assigner does not take the low-res logits."""
        a, Np, T = self.mask_assigner[stage], self.num_proposals, self.num_thing_classes
        n = (masks if masks is not None else lowres[0]).shape[0]
        c = [cls[i][:Np, :T] if cls is not None else None for i in range(n)]
        if lowres is not None and hasattr(a, 'assign_batch_lowres'):
            lows = [lowres[0][i][:Np] for i in range(n)]
            if a.lowres_ready(lows, lowres[1], c, gt_masks, gt_labels):
                return a.assign_batch_lowres(lows, lowres[1], c, gt_masks, gt_labels)
        if masks is None:
            masks = ops.upsample_bilinear(lowres[0], lowres[1]) if lowres[0].is_cuda and lowres[0].dtype == torch.float32 else \
                torch.nn.functional.interpolate(lowres[0], scale_factor=lowres[1], mode='bilinear', align_corners=False)
        m = [masks[i][:Np] for i in range(n)]
        if hasattr(a, 'assign_batch'):
            return a.assign_batch(m, c, gt_masks, gt_labels, img_metas)
        return [a.assign(m[i], c[i], gt_masks[i], gt_labels[i], img_meta=img_metas[i]) for i in range(n)]

    def forward_train(self, x, proposal_feats, mask_preds, cls_score, img_metas, gt_masks, gt_labels, gt_bboxes_ignore=None,
                      imgs_whwh=None, gt_bboxes=None, gt_sem_seg=None, gt_sem_cls=None):
        """-> dict of `s{stage}_loss_cls / loss_mask / loss_dice / loss_rank / pos_acc`          reference :139-231"""
        return self._train_stages(x, proposal_feats, mask_preds, cls_score, img_metas, gt_masks, gt_labels, imgs_whwh,
                                  gt_sem_seg, gt_sem_cls)[0]

    # ---- post-head pipeline (reference :233-283, 332-370, 467-524)
    @staticmethod
    def _cfg(cfg, key):
        return cfg[key] if isinstance(cfg, dict) else getattr(cfg, key)

    @staticmethod
    def _meta_geometry(meta):
        return tuple(meta['img_shape'][:2]), tuple(meta['batch_input_shape'][:2]), tuple(meta['ori_shape'][:2])

    def _panoptic_joint(self, cls_score, mask_logits, test_cfg, img_meta, upsample_stride, want_bbox=False):
        """Frames [B] sharing one img_meta -> device tensors (panoptic_seg [B,Ho,Wo] int32, info [B,K,6], nseg [B][, bbox])."""
        if not self.merge_joint:
            raise RuntimeError('_panoptic_joint is the merge_joint=True pipeline; merge_joint=False goes through get_panoptic')
        merge_cfg = self._cfg(test_cfg, 'merge_stuff_thing')
        img, bis, ori = self._meta_geometry(img_meta)
        return ops.panoptic_joint(cls_score, mask_logits, self.num_proposals, self.num_thing_classes,
                                  self._cfg(test_cfg, 'max_per_img'), self._cfg(merge_cfg, 'instance_score_thr'),
                                  self._cfg(merge_cfg, 'overlap_thr'), img, bis, ori, upsample_stride=upsample_stride,
                                  want_bbox=want_bbox)

    def _segments_info(self, info_b):
        """info [K,6] (host numpy) -> the reference's segments_info list, in segment-id order (reference :505-521)."""
        out = []
        acc = np.nonzero(info_b[:, 2] > 0)[0]
        for k in acc[np.argsort(info_b[acc, 2], kind='stable')]:
            label, sid = int(info_b[k, 1]), int(info_b[k, 2])
            if label < self.num_thing_classes:
                out.append(dict(id=sid, isthing=True, score=float(info_b[k, 5:6].view(np.float32)[0]), category_id=label,
                                instance_id=int(k)))
            else:
                out.append(dict(id=sid, isthing=False, category_id=label - self.num_thing_classes + 1, area=int(info_b[k, 3])))
        return out

    def things_for_tracking(self, info_b, bbox_b):
        """Host arrays of one frame (info [K,6], bbox [K,4]) -> what the video detector hands its tracker
        (`get_things_id_for_tracking` + `tensor_mask2box`, knet/video/knet_quansi_dense_embed_fc_joint_train.py:541-584,
        673-685): (instance ids, thing labels, boxes [n,4] (xmin, ymin, xmax, ymax), scores) in segment order."""
        acc = np.nonzero((info_b[:, 2] > 0) & (info_b[:, 1] < self.num_thing_classes))[0]
        acc = acc[np.argsort(info_b[acc, 2], kind='stable')]
        return (acc.tolist(), info_b[acc, 1].tolist(), bbox_b[acc].astype(np.float32),
                info_b[acc, 5:6].copy().view(np.float32)[:, 0].tolist())

    # ---- thing-first merge (merge_joint=False): reference :385-465
    def merge_stuff_thing(self, thing_masks, thing_labels, thing_scores, stuff_masks, stuff_labels, stuff_scores, merge_cfg=None):
        """Boolean masks pasted in score order, things first (reference :385-465) — on the device (`vkn_panoptic_thing_first_u8`),
        one D2H copy of the map and a [K, 5] table.  -> (panoptic_seg int32 ndarray, segments_info)."""
        dev = thing_masks.device
        thing_order = torch.argsort(-thing_scores)
        sorted_inds = torch.argsort(-stuff_scores)
        lab_sorted = stuff_labels[sorted_inds].cpu().tolist()
        if len(set(lab_sorted)) == len(lab_sorted):
            sm, sl, so = stuff_masks, stuff_labels, sorted_inds
        else:   # several masks per label: OR them, one entry per distinct label in first-occurrence (score) order (:440-448)
            uniq = list(dict.fromkeys(lab_sorted))
            sm = torch.stack([stuff_masks[stuff_labels == u].bool().any(0) for u in uniq])
            sl = torch.tensor(uniq, device=dev)
            so = torch.arange(len(uniq), device=dev)
        seg, info, nseg = ops.panoptic_thing_first(thing_masks, thing_scores, thing_labels, thing_order, sm, sl, so,
                                                   self._cfg(merge_cfg, 'instance_score_thr'), self._cfg(merge_cfg, 'iou_thr'),
                                                   self._cfg(merge_cfg, 'stuff_max_area'))
        info_h = info.cpu().numpy()
        segments_info = []
        for row in info_h[info_h[:, 0] > 0]:
            if row[1] == 0:
                segments_info.append(dict(id=int(row[0]), isthing=True, score=float(row[4:5].view(np.float32)[0]), category_id=int(row[2]),
                                          instance_id=int(row[3])))
            else:
                segments_info.append(dict(id=int(row[0]), isthing=False, category_id=int(row[2]), area=int(row[3])))
        return seg.cpu().numpy(), segments_info

    # ---- the reference's merge helpers under their own names, for callers that hold MASKS (the fused `get_panoptic` path above
    #      goes from class scores + logits to the map in one kernel and never materialises them)
    def split_thing_stuff(self, mask_preds, det_labels, cls_scores):
        """Rows [0, num_proposals) are things, the rest stuff with labels renumbered from 1 (reference :372-384)."""
        n, first_stuff = self.num_proposals, self.num_thing_classes - 1
        return (mask_preds[:n], det_labels[:n], cls_scores[:n], mask_preds[n:], det_labels[n:] - first_stuff, cls_scores[n:])

    def _joint_merge(self, total_masks, total_labels, total_scores, merge_cfg):
        """The joint merge rule (reference :467-524) without its per-mask loop: a pixel belongs to the mask with the largest
        score * probability; a mask survives if it wins pixels, covers any at p >= 0.5, keeps at least `overlap_thr` of that
        area, and — for things — scores at least `instance_score_thr`; survivors are numbered 1.. in descending score order.
        -> (winner [H, W], segment id per mask [K] (0 = dropped), won pixels per mask [K])."""
        K = total_masks.shape[0]
        winner = (total_scores.view(-1, 1, 1) * total_masks).argmax(0)
        area = torch.bincount(winner.flatten(), minlength=K)
        solid = (total_masks >= 0.5).flatten(1).sum(1)
        keep = (area > 0) & (solid > 0) & (area.double() / solid.clamp(min=1).double() >= float(self._cfg(merge_cfg, 'overlap_thr')))
        keep &= ~((total_labels < self.num_thing_classes) & (total_scores < self._cfg(merge_cfg, 'instance_score_thr')))
        order = torch.argsort(-total_scores, stable=True)
        kept_in_order = keep[order]
        seg_id = torch.zeros(K, dtype=torch.long, device=total_masks.device)
        seg_id[order] = torch.where(kept_in_order, torch.cumsum(kept_in_order.long(), 0), seg_id)
        return winner, seg_id, area

    def _joint_segments(self, seg_id, total_labels, total_scores, area):
        """segments_info in segment order + the mask index of every accepted thing, from ONE device -> host copy."""
        table = torch.stack([seg_id, total_labels.long(), area]).cpu().numpy()
        scores = total_scores.detach().cpu().numpy()
        T = self.num_thing_classes
        rows = [k for k in np.argsort(table[0], kind='stable') if table[0][k] > 0]
        info, things = [], []
        for k in rows:
            sid, lab = int(table[0][k]), int(table[1][k])
            if lab < T:
                info.append(dict(id=sid, isthing=True, score=float(scores[k]), category_id=lab, instance_id=int(k)))
                things.append(int(k))
            else:
                info.append(dict(id=sid, isthing=False, category_id=lab - T + 1, area=int(table[2][k])))
        return info, things

    def merge_stuff_thing_stuff_joint(self, thing_masks, thing_labels, thing_scores, stuff_masks, stuff_labels, stuff_scores,
                                      merge_cfg=None):
        """-> (panoptic_seg int32 ndarray, segments_info) from SOFT masks (probabilities), things and stuff competing per pixel."""
        masks, labels = torch.cat([thing_masks, stuff_masks]), torch.cat([thing_labels, stuff_labels])
        scores = torch.cat([thing_scores, stuff_scores])
        winner, seg_id, area = self._joint_merge(masks, labels, scores, merge_cfg)
        info, _ = self._joint_segments(seg_id, labels, scores, area)
        return seg_id[winner].to(torch.int32).cpu().numpy(), info

    def _get_panoptic_thing_first(self, cls_scores, mask_preds, test_cfg, img_meta):
        """reference :332-370 with merge_joint=False: top-k things and score-sorted stuff, rescaled and thresholded, then merged."""
        Np, T = self.num_proposals, self.num_thing_classes
        last = self.mask_head[-1]
        thing_scores, topk = cls_scores[:Np][:, :T].flatten(0, 1).topk(self._cfg(self.test_cfg, 'max_per_img'), sorted=True)
        mask_indices, thing_labels = topk // T, topk % T
        thr = self._cfg(test_cfg, 'mask_thr')
        thing_masks = last.rescale_masks(mask_preds[:Np][mask_indices], img_meta) > thr
        bbox_result, segm_result = last.segm2result(thing_masks, thing_labels, thing_scores)
        stuff_scores, stuff_inds = torch.sort(cls_scores[Np:][:, T:].diag(), descending=True)
        stuff_masks = last.rescale_masks(mask_preds[Np:][stuff_inds], img_meta) > thr
        pan = self.merge_stuff_thing(thing_masks, thing_labels, thing_scores, stuff_masks, stuff_inds + 1, stuff_scores,
                                     self._cfg(test_cfg, 'merge_stuff_thing'))
        return bbox_result, segm_result, pan

    def get_panoptic(self, cls_scores, mask_preds, test_cfg, img_meta):
        """One image, the reference's signature (:332-370): `mask_preds` are the (already up-scaled) `scaled_mask_preds[img]`.
        Returns `(bbox_result, segm_result, (panoptic_seg int32 ndarray, segments_info))`; the first two are None: they are
        the K full-resolution thing masks as host arrays (`segm2result`), which this pipeline exists to NOT materialise."""
        if not self.merge_joint:
            return self._get_panoptic_thing_first(cls_scores, mask_preds, test_cfg, img_meta)
        seg, info, nseg = self._panoptic_joint(cls_scores[None], mask_preds[None], test_cfg, img_meta, 1)
        info_h = info[0].cpu().numpy()
        if int(nseg[0]) < 0:
            raise RuntimeError('vkn_panoptic_joint_f32: internal LDS capacity error')
        return None, None, (seg[0].cpu().numpy(), self._segments_info(info_h))

    def _panoptic_results(self, cls_score, mask_preds, img_metas, upsample_stride):
        """All frames of a batch: one fused call per group of frames with identical geometry (normally one group)."""
        groups = {}
        for i, m in enumerate(img_metas):
            groups.setdefault(self._meta_geometry(m), []).append(i)
        results = [None] * len(img_metas)
        for idx in groups.values():
            sel = torch.as_tensor(idx, device=cls_score.device)
            whole = len(idx) == len(img_metas)
            seg, info, nseg = self._panoptic_joint(cls_score if whole else cls_score[sel], mask_preds if whole else mask_preds[sel],
                                                   self.test_cfg, img_metas[idx[0]], upsample_stride)
            seg_h, info_h, nseg_h = seg.cpu().numpy(), info.cpu().numpy(), nseg.cpu().numpy()   # the only host sync
            if (nseg_h < 0).any():
                raise RuntimeError('vkn_panoptic_joint_f32: internal LDS capacity error')
            for j, i in enumerate(idx):
                results[i] = (seg_h[j], self._segments_info(info_h[j]), info_h[j])
        return results

    def simple_test(self, x, proposal_feats, mask_preds, cls_score, img_metas, imgs_whwh=None, rescale=False):
        """Stage loop + panoptic results per image (reference :233-283): a list of
        `(bbox_result, segm_result, (panoptic_seg, segments_info))` with bbox/segm results None (see get_panoptic)."""
        if not self._fused_ok(x):
            raise NotImplementedError('simple_test needs the fused GPU head (eval mode, CUDA tensors)')
        if not self.do_panoptic:
            # instance-only results (reference :270-281): top-`max_per_img` (kernel, class) pairs -> their up-scaled masks ->
            # rescale / threshold (`get_seg_masks`).  Selection and resampling run on the device, one D2H copy of K bool masks.
            _, cls, _, scaled, _ = self._head_forward(x, proposal_feats, mask_preds)
            return [self._instance_result(cls[i], scaled[i], img_metas[i]) for i in range(len(img_metas))]
        if not self.merge_joint:   # thing-first merge: needs the K rescaled boolean masks (reference :263-269, 332-370)
            _, cls, _, scaled, _ = self._head_forward(x, proposal_feats, mask_preds)
            return [self.get_panoptic(cls[i], scaled[i], self.test_cfg, img_metas[i]) for i in range(len(img_metas))]
        _, cls, masks, _, _ = self._head_forward(x, proposal_feats, mask_preds, want_scaled=False)
        up = self.mask_head[-1].mask_upsample_stride
        return [(None, None, (seg, info)) for seg, info, _ in self._panoptic_results(cls, masks, img_metas, up)]

    def _instance_topk(self, cls_score_per_img):
        num_classes = self.mask_head[-1].num_classes
        scores, topk = cls_score_per_img.flatten(0, 1).topk(self._cfg(self.test_cfg, 'max_per_img'), sorted=True)
        return scores, topk // num_classes, topk % num_classes

    def _instance_result(self, cls_score_per_img, scaled_mask_preds_per_img, img_meta):
        scores, mask_indices, labels = self._instance_topk(cls_score_per_img)
        return self.mask_head[-1].get_seg_masks(scaled_mask_preds_per_img[mask_indices], labels, scores, self.test_cfg, img_meta)

    def aug_test(self, features, proposal_list, img_metas, rescale=False):
        raise NotImplementedError('SparseMask does not support `aug_test`')


@register_head
class VideoKernelIterHead(KernelIterHead):
    """knet/video/kernel_iter_head.py:10-564 (with_track + previous-frame link in the last stage only)."""

    def _init_extra(self, kwargs):
        self.with_track = kwargs.pop('with_track', False)

    def _mask_forward(self, stage, x, object_feats, mask_preds, img_metas, previous_obj_feats=None,
                      previous_mask_preds=None, previous_x_feats=None):
        mask_head = self.mask_head[stage]
        cls_score, mask_preds, object_feats, x_feats, object_feats_track = mask_head(
            x, object_feats, mask_preds, img_metas=img_metas, previous_obj_feats=previous_obj_feats,
            previous_mask_preds=previous_mask_preds, previous_x_feats=previous_x_feats)
        if mask_head.mask_upsample_stride > 1 and (stage == self.num_stages - 1 or self.training):
            scaled_mask_preds = self._upsample(mask_preds, mask_head.mask_upsample_stride)
        else:
            scaled_mask_preds = mask_preds
        return dict(cls_score=cls_score, mask_preds=mask_preds, scaled_mask_preds=scaled_mask_preds,
                    object_feats=object_feats, object_feats_track=object_feats_track, x_feats=x_feats)

    def forward_train(self, x, proposal_feats, mask_preds, cls_score, img_metas, gt_masks, gt_labels, gt_pids=None,
                      gt_bboxes_ignore=None, imgs_whwh=None, gt_bboxes=None, gt_sem_seg=None, gt_sem_cls=None):
        """knet/video/kernel_iter_head.py:150-253: losses, and with `with_track` also the last stage's outputs."""
        losses, r = self._train_stages(x, proposal_feats, mask_preds, cls_score, img_metas, gt_masks, gt_labels, imgs_whwh,
                                       gt_sem_seg, gt_sem_cls)
        if self.with_track:
            return losses, r['object_feats'], r['cls_score'], r['mask_preds'], r['scaled_mask_preds']
        return losses

    def forward_train_with_previous(self, x, proposal_feats, mask_preds, cls_score, img_metas, gt_masks, gt_labels, gt_pids=None,
                                    gt_bboxes_ignore=None, imgs_whwh=None, gt_bboxes=None, gt_sem_seg=None, gt_sem_cls=None,
                                    previous_obj_feats=None, previous_mask_preds=None, previous_x_feats=None):
        """knet/video/kernel_iter_head.py:255-376: the previous frame's kernels reach the LAST stage only (:300-303)."""
        last = self.num_stages - 1

        def kw(stage):
            on = stage == last
            return dict(previous_obj_feats=previous_obj_feats if on else None,
                        previous_mask_preds=previous_mask_preds if on else None,
                        previous_x_feats=previous_x_feats if on else None)

        losses, r = self._train_stages(x, proposal_feats, mask_preds, cls_score, img_metas, gt_masks, gt_labels, imgs_whwh,
                                       gt_sem_seg, gt_sem_cls, stage_kwargs=kw)
        if self.with_track:
            return (losses, r['object_feats'], r['cls_score'], r['mask_preds'], r['scaled_mask_preds'],
                    r['object_feats_track'])
        return losses

    def simple_test_mask_preds(self, x, proposal_feats, mask_preds, cls_score, img_metas):
        """reference :508-527"""
        return self.simple_test_mask_preds_plus_previous(x, proposal_feats, mask_preds, cls_score, img_metas)

    def simple_test_mask_preds_plus_previous(self, x, proposal_feats, mask_preds, cls_score, img_metas,
                                             previous_obj_feats=None, previous_mask_preds=None, previous_x_feats=None,
                                             return_track=False):
        """-> (object_feats, cls_score.sigmoid(), mask_preds, scaled_mask_preds)                   reference :529-564
        `previous_*` reach the LAST stage only (:544-546).  `return_track=True` (an extension: the reference computes the
        tracking embedding and drops it here) appends object_feats_track."""
        if self._fused_ok(x) and all(isinstance(h, VideoKernelUpdateHead) for h in self.mask_head):
            last = self.mask_head[-1]
            link = previous_obj_feats is not None and last.previous is not None
            out = self._head_forward(x, proposal_feats, mask_preds, previous_obj_feats if link else None,
                                     want_track=link and last.previous_type is not None)
            return out if return_track else out[:4]
        object_feats, track = proposal_feats, None
        for stage in range(self.num_stages):
            last = stage == self.num_stages - 1
            r = self._mask_forward(stage, x, object_feats, mask_preds, img_metas,
                                   previous_obj_feats=previous_obj_feats if last else None,
                                   previous_mask_preds=previous_mask_preds if last else None,
                                   previous_x_feats=previous_x_feats if last else None)
            object_feats, cls_score = r['object_feats'], r['cls_score']
            mask_preds, scaled_mask_preds, track = r['mask_preds'], r['scaled_mask_preds'], r['object_feats_track']
        cls_score = cls_score.sigmoid() if self.mask_head[-1].loss_cls.use_sigmoid else cls_score.softmax(-1)[..., :-1]
        out = (object_feats, cls_score, mask_preds, scaled_mask_preds)
        return out + (track,) if return_track else out

    def clip_forward(self, x, proposal_feats, mask_preds, first_previous_obj_feats=None, want_scaled=True):
        """Clip-batched inference (an MI355X extension; the reference walks a video one frame per call,
        knet/video/knet_quansi_dense_embed_fc_joint_train.py:472-612).  x [T,C,H,W] are T CONSECUTIVE frames.  Masks, cls and
        kernels of frame t do not depend on frame t-1, so all T frames run as one batch; only the tracking
        embedding does: track[t] = link(cur = obj[t], prev = obj[t-1]) with obj[-1] = `first_previous_obj_feats`
        (None = first frame of the video: the reference then uses object_feats as the tracking feature, :474-475).
        Heads with `previous_link` ("update" configs): the LAST stage's incoming kernels of frame t are rewritten from frame t-1's
        final kernels, so masks DO depend on the previous frame — the library then runs the last stage's [N x C] chain frame by
        frame between the batched last gather and the batched last decode (vkn_head_forward_link_f32); the first frame of a video
        (no previous kernels) runs without the link, exactly like the reference's first `simple_test_with_previous` call.
        -> (object_feats [T,N,C,1,1], cls [T,N,ncls], mask_preds, scaled_mask_preds, object_feats_track [T,N,C,1,1])"""
        last = self.mask_head[-1]
        in_call_only = getattr(last, 'previous', None) is not None and (
            getattr(last, 'previous_link', None) is not None or getattr(last, 'previous_type', None) in ('update', 'update_obj'))
        if in_call_only and first_previous_obj_feats is None:
            # (previous_link rewrites the last stage's kernels; the "update" / "update_obj" tracking links need the stage's x_feat /
            #  their own updator: both exist only inside the head call)
            # frame 0 has nothing to link to; frames 1.. form a clip whose first previous kernels are frame 0's
            o0, c0, m0, s0, _ = self._head_forward(x[:1], proposal_feats[:1], mask_preds[:1], want_scaled=want_scaled)
            if x.shape[0] == 1:
                return o0, c0, m0, s0, o0
            o1, c1, m1, s1, t1 = self._head_forward(x[1:], proposal_feats[1:], mask_preds[1:], want_scaled=want_scaled,
                                                    clip_first_prev=o0)
            return (torch.cat([o0, o1]), torch.cat([c0, c1]), torch.cat([m0, m1]), torch.cat([s0, s1]), torch.cat([o0, t1]))
        if first_previous_obj_feats is not None and getattr(last, 'previous', None) is not None:
            # the whole clip, link included, in one C call (VKN_FLAG_CLIP_LINK)
            obj, cls, masks, scaled, track = self._head_forward(x, proposal_feats, mask_preds, want_scaled=want_scaled,
                                                                clip_first_prev=first_previous_obj_feats)
            return obj, cls, masks, scaled, track
        obj, cls, masks, scaled, _ = self._head_forward(x, proposal_feats, mask_preds)
        T, N = obj.shape[:2]
        C = last.in_channels
        cur = obj.reshape(T, N, C)
        if getattr(last, 'previous', None) is None:
            return obj, cls, masks, scaled, obj
        prev0 = cur[:1] if first_previous_obj_feats is None else first_previous_obj_feats.reshape(1, N, C)
        prev = torch.cat([prev0, cur[:-1]], dim=0)
        track = ops.track_link(last.make_dims(T, N, x.shape[-2], x.shape[-1]), last.stage_pack(x.device), cur, prev)
        if first_previous_obj_feats is None:
            track[0] = cur[0]
        return obj, cls, masks, scaled, track.reshape(obj.shape)

    def linked_block_phases(self, x, proposal_feats, mask_preds, want_scaled=True):
        """`run_phase` callable for `dist.linked_block_forward`: this head on one rank's contiguous block of a clip, in the three phases
        of `vkn_head_forward_link_f32` (VKN_FLAG_PHASE_A / B / C).  Only for heads with a `previous_link` block (the others shard
        without phases).  Phase 'B' returns the block's final kernels [T, N, C]; phase 'C' the five outputs of `clip_forward`."""
        last = self.mask_head[-1]
        if getattr(last, 'previous', None) is None or getattr(last, 'previous_link', None) is None:
            raise ValueError('linked_block_phases: the last stage has no previous_link block — shard the clip with clip_forward + '
                             'dist.neighbour_last_kernels instead')
        h0 = self.mask_head[0]
        T, N = proposal_feats.shape[:2]
        C, K = h0.in_channels, h0.conv_kernel_size
        dims = h0.make_dims(T, N, x.shape[-2], x.shape[-1])
        packs = [h.stage_pack(x.device) for h in self.mask_head]
        link_pre, link_track, track_src = last.link_packs(x.device)
        pf = proposal_feats.reshape(T, N, C)
        state = {}

        def run(name, prev):
            bit = {'A': ops.PHASE_A, 'B': ops.PHASE_B, 'C': ops.PHASE_C}[name]
            p = prev if prev is not None else pf.new_zeros(1, N, C)      # (phase A never reads it: no cross-frame input yet)
            # the gather sums and stage S-2's kernels live in this (thread, device, stream)'s workspace between the phases: a phase on
            # another stream, or a larger call in between that re-allocated the buffer, would read garbage — loud instead
            key = (torch.cuda.current_stream(x.device).cuda_stream, ops.workspace_generation(x.device))
            if name != 'A' and state.get('ws_key') not in (None, key):
                raise RuntimeError('linked_block_phases: the workspace that holds the state of phase A was re-allocated (or the stream '
                                   'changed) before phase %s — run the three phases of a block on one stream, with no larger call between them' % name)
            state['out'] = ops.head_forward(dims, packs, x, pf, mask_preds, None, last.mask_upsample_stride, want_scaled=want_scaled,
                                            flags=getattr(h0, 'vkn_flags', 0), clip_first_prev=p.reshape(1, N, C), link_pre=link_pre,
                                            link_track=link_track, track_src=track_src, phase=bit, out=state.get('out'))
            obj, cls, masks, scaled, track = state['out']
            state['ws_key'] = (torch.cuda.current_stream(x.device).cuda_stream, ops.workspace_generation(x.device))
            if name == 'B':
                return obj
            if name == 'C':
                return obj.reshape(T, N, C, K, K), cls, masks, scaled, track.reshape(T, N, C, K, K)
            return None
        return run

    def get_masked_feature(self, x, mask_pred):
        """`einsum('bnhw,bchw->bnc', (sigmoid(mask_pred) > 0.5).float(), x)` (knet/video/kernel_iter_head.py:566-571): the HIP gather."""
        return ops.mask_gather(x, mask_pred, 0.5)[0]

    def merge_stuff_thing_stuff_joint(self, thing_masks, thing_labels, thing_scores, stuff_masks, stuff_labels, stuff_scores,
                                      merge_cfg=None, thing_obj=None, stuff_obj=None):
        """The joint merge plus the tracking embeddings of the accepted things in segment order (:832-905)."""
        masks, labels = torch.cat([thing_masks, stuff_masks]), torch.cat([thing_labels, stuff_labels])
        scores = torch.cat([thing_scores, stuff_scores])
        winner, seg_id, area = self._joint_merge(masks, labels, scores, merge_cfg)
        info, things = self._joint_segments(seg_id, labels, scores, area)
        feats = None
        if thing_obj is not None:
            obj = torch.cat([thing_obj, stuff_obj]) if stuff_obj is not None else thing_obj
            feats = obj[torch.as_tensor(things, dtype=torch.long, device=obj.device)]
        return (seg_id[winner].to(torch.int32).cpu().numpy(), info), feats

    def merge_stuff_thing_thing_first(self, thing_masks, thing_labels, thing_scores, stuff_masks, stuff_labels, stuff_scores,
                                      merge_cfg=None, thing_obj_feat=None, stuff_obj_feat=None):
        """Things pasted first in score order, then stuff (:656-742) = the image head's `merge_stuff_thing`, plus the embeddings of
        the accepted things, which the reference indexes in score-sorted order."""
        pan = self.merge_stuff_thing(thing_masks, thing_labels, thing_scores, stuff_masks, stuff_labels, stuff_scores, merge_cfg)
        feats = None
        if thing_obj_feat is not None:
            ids = [s_['instance_id'] for s_ in pan[1] if s_['isthing']]
            feats = thing_obj_feat[torch.argsort(-thing_scores)][torch.as_tensor(ids, dtype=torch.long, device=thing_obj_feat.device)]
        return pan, feats

    def get_panoptic(self, cls_scores, mask_preds, test_cfg, img_meta, obj_feat=None):
        """Video signature (knet/video/kernel_iter_head.py:591-640): 5-tuple
        `(bbox_result, segm_result, thing_mask_preds, panoptic_result, thing_obj_feat)`.  The first three (boxes and the K
        full-resolution thing masks for the tracker) are None — see KernelIterHead.get_panoptic; `thing_obj_feat` are the
        tracking embeddings of the accepted thing segments in segment order (`sort_obj_fea[things_ids]`, :903)."""
        if not self.merge_joint:
            # merge_stuff_thing_thing_first (:656-742): the det head's thing-first merge + the accepted things' embeddings, which
            # the reference indexes in SCORE-SORTED order: thing_obj_feat[sorted_inds][instance_ids] (:676, :742)
            Np, T = self.num_proposals, self.num_thing_classes
            bbox_result, segm_result, pan = self._get_panoptic_thing_first(cls_scores, mask_preds, test_cfg, img_meta)
            tfeat = None
            if obj_feat is not None:
                thing_scores, topk = cls_scores[:Np][:, :T].flatten(0, 1).topk(self._cfg(self.test_cfg, 'max_per_img'), sorted=True)
                sel = obj_feat[:Np][topk // T][torch.argsort(-thing_scores)]
                ids = [s_['instance_id'] for s_ in pan[1] if s_['isthing']]
                tfeat = sel[torch.as_tensor(ids, dtype=torch.long, device=sel.device)]
            return bbox_result, segm_result, None, pan, tfeat
        seg, info, nseg = self._panoptic_joint(cls_scores[None], mask_preds[None], test_cfg, img_meta, 1)
        info_h = info[0].cpu().numpy()
        if int(nseg[0]) < 0:
            raise RuntimeError('vkn_panoptic_joint_f32: internal LDS capacity error')
        return None, None, None, (seg[0].cpu().numpy(), self._segments_info(info_h)), self._thing_obj_feat(info_h, obj_feat)

    def _thing_obj_feat(self, info_b, obj_feat):
        if obj_feat is None:
            return None
        acc = np.nonzero((info_b[:, 2] > 0) & (info_b[:, 1] < self.num_thing_classes))[0]
        acc = acc[np.argsort(info_b[acc, 2], kind='stable')]
        rows = torch.as_tensor(info_b[acc, 0].astype(np.int64), device=obj_feat.device)
        return obj_feat[rows]

    def simple_test_with_previous(self, x, proposal_feats, mask_preds, cls_score, img_metas, previous_obj_feats=None,
                                  previous_mask_preds=None, previous_x_feats=None, is_first=False):
        """Stage loop with the last-stage tracking link + panoptic results per image
        (knet/video/kernel_iter_head.py:435-506).  `results[i]` is the 5-tuple of `get_panoptic`; with `with_track` the
        reference's `(results, object_feats, cls_score, mask_preds, scaled_mask_preds)` is returned."""
        if not self.do_panoptic:
            raise NotImplementedError('the video head is panoptic-only in every shipped config (do_panoptic / merge_joint)')
        if not (self._fused_ok(x) and all(isinstance(h, VideoKernelUpdateHead) for h in self.mask_head)):
            raise NotImplementedError('simple_test_with_previous needs the fused GPU head (eval mode, CUDA tensors)')
        last = self.mask_head[-1]
        link = previous_obj_feats is not None and last.previous is not None
        obj, cls, masks, scaled, track = self._head_forward(x, proposal_feats, mask_preds,
                                                            previous_obj_feats if link else None,
                                                            want_track=link and last.previous_type is not None,
                                                            want_scaled=self.with_track)
        if is_first or track is None:
            track = obj                                                                                   # :474-475
        up = self.mask_head[-1].mask_upsample_stride
        results = []
        for i, (seg, seg_info, info_h) in enumerate(self._panoptic_results(cls, masks, img_metas, up)):
            results.append((None, None, None, (seg, seg_info), self._thing_obj_feat(info_h, track[i])))
        if self.with_track:
            return results, obj, cls, masks, scaled
        return results

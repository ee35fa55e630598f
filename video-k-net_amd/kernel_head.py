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

from . import ops
from .registry import HAVE_MM, HEADS, register_head


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
        self._init_layers()

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
        cat = self.cat_stuff_mask and not self.training
        if self.cat_stuff_mask and self.training:
            raise NotImplementedError('training is a later row (SURVEY.md §8(f)); this build covers inference')
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

    def forward_train(self, *args, **kwargs):
        raise NotImplementedError('training (assignment + losses, knet/det/kernel_head.py:265-504) is a later row; this build '
                                  'covers the inference hot path')

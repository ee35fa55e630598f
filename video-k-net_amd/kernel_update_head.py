"""`KernelUpdateHead` / `VideoKernelUpdateHead` — drop-ins for one refinement stage of the reference
(knet/det/kernel_update_head.py:16-277, knet/video/kernel_update_head.py:17-541): same `HEADS` registration, same ctor kwargs,
same module tree (=> the same 44 (+16 video) state-dict keys per stage), same call signature and returns.
The forward arithmetic is one C-ABI call into libvkn.so (gather -> update -> decode HIP kernels).

Sub-modules whose arithmetic lives in third-party mmcv in the reference (`MultiheadAttention`, `FFN`, `ConvModule`) are
parameter containers here with the same attribute nesting, so the key names match (`attention.attn.in_proj_weight`,
`ffn.layers.0.0.weight`, `feat_transform.conv.weight`, ...).
"""
import math

import torch
import torch.nn as nn

from . import autograd as vag
from . import chain_train
from . import ops
from .losses import accuracy, reduce_mean
from .registry import build_loss, build_transformer_layer, register_head


def _updator_torch(ku, update_feature, input_feature):
    """The adaptive kernel update on torch ops — only for `_chain_autograd` (the A/B arm of the device chain and the fallback for shapes
    the library's training kernels decline, `chain_train.supported`) and the CPU-side tests of the gradient reducer.  Laid out the way
    `chain_train.UpdatorCoreFn` launches it: the dynamic / input projections [rows, 2C] = (gate half | value half), BOTH gate layers
    as one GEMM on the stacked weights, then the gated mix and the output projection.  [M, C] (or [B, N, C]) x [M, K*K, C] -> [M, K*K, C]."""
    F = torch.nn.functional
    C = ku.feat_channels
    u = update_feature.reshape(-1, ku.in_channels)
    M = u.shape[0]
    dyn = F.linear(u, ku.dynamic_layer.weight, ku.dynamic_layer.bias).unsqueeze(1)                       # [M, 1, 2C]
    inp = F.linear(input_feature.reshape(M, -1, C), ku.input_layer.weight, ku.input_layer.bias)          # [M, K*K, 2C]
    gates = F.linear(inp[..., :C] * dyn[..., :C], torch.cat([ku.input_gate.weight, ku.update_gate.weight]),
                     torch.cat([ku.input_gate.bias, ku.update_gate.bias]))                               # [M, K*K, 2C]
    keep_input = torch.sigmoid(ku.input_norm_in(gates[..., :C]))
    take_update = torch.sigmoid(ku.norm_in(gates[..., C:]))
    mixed = take_update * ku.norm_out(dyn[..., C:]) + keep_input * ku.input_norm_out(inp[..., C:])
    return torch.relu(ku.fc_norm(ku.fc_layer(mixed)))


class _MHAParams(nn.Module):
    """mmcv `MultiheadAttention(embed_dims, num_heads, attn_drop)` as a parameter container: `.attn` is a real
    `nn.MultiheadAttention` so parameter names, shapes and default init are torch's own."""

    def __init__(self, embed_dims, num_heads, dropout=0.0):
        super().__init__()
        if dropout != 0.0:
            raise NotImplementedError('dropout must be 0.0 (every shipped config)')
        self.embed_dims, self.num_heads = embed_dims, num_heads
        self.attn = nn.MultiheadAttention(embed_dims, num_heads, dropout)


class _FFNParams(nn.Module):
    """mmcv `FFN(embed_dims, feedforward_channels, num_fcs, act_cfg, dropout)` as a parameter container
    (`layers = Seq(Seq(Linear, ReLU, Dropout) x (num_fcs-1), Linear, Dropout)`)."""

    def __init__(self, embed_dims, feedforward_channels, num_fcs=2, dropout=0.0):
        super().__init__()
        if num_fcs != 2:
            raise NotImplementedError('num_ffn_fcs must be 2 (every shipped config)')
        if dropout != 0.0:
            raise NotImplementedError('dropout must be 0.0 (every shipped config)')
        self.layers = nn.Sequential(
            nn.Sequential(nn.Linear(embed_dims, feedforward_channels), nn.ReLU(inplace=True), nn.Dropout(dropout)),
            nn.Linear(feedforward_channels, embed_dims), nn.Dropout(dropout))


class _ConvParams(nn.Module):
    """mmcv `ConvModule(C, C, 1, conv_cfg=Conv2d, act_cfg=None)`: a biased 1x1 conv, no norm, no activation."""

    def __init__(self, channels, kernel_size):
        super().__init__()
        if kernel_size != 1:
            raise NotImplementedError('feat_transform kernel_size must be 1 (every shipped config)')
        self.conv = nn.Conv2d(channels, channels, 1)


class _ChainGraphModule(nn.Module):
    """The [B*N, C] chain of one stage (`KernelUpdateHead._chain_autograd`) as a callable for `torch.cuda.make_graphed_callables`:
    its forward and backward become TWO hipGraph launches instead of ~150 + ~300 eager torch launches per stage.  The chain's shapes
    are static ([B, N, C] whatever the ground truth looks like), unlike the losses behind it, so this is the part of a training
    step that can be captured.  The head is held WITHOUT registering it as a sub-module (the head owns this object); `parameters()`
    hands the head's parameters to the capture so that their gradients are part of the backward graph."""

    def __init__(self, head, has_prev):
        super().__init__()
        object.__setattr__(self, 'head', head)
        self.has_prev = has_prev
        self.pattern = None          # which of the five outputs exist (cls_score / bias / track may be None)

    def parameters(self, recurse=True):
        return iter(list(self.head.parameters()))

    def forward(self, x_feat, proposal_feat, prev=None):
        out = self.head._chain_impl(x_feat, proposal_feat, prev if self.has_prev else None)
        self.pattern = tuple(o is not None for o in out)
        return tuple(o for o in out if o is not None)


class _ChainGraphRunner:
    """One captured (forward, backward) hipGraph pair of a stage's chain for fixed shapes.

    `torch.cuda.make_graphed_callables` would do the capture too, but it returns every parameter gradient as an autograd output:
    ~150 `detach` + accumulation nodes + per-parameter hooks per stage and backward — a third of the host time of a step that is
    host-bound.  Here the parameters are constants of the captured graphs as far as autograd is concerned: the backward graph
    computes their gradients into static buffers and `_deliver` hands them over in bulk — `p.grad = buffer` when the parameter has
    no gradient yet (the buffer is consumed before the next replay: optimizer step / bucket copy), ONE multi-tensor add otherwise —
    and then tells whoever registered `head.on_param_grads` (the bucketed all-reducer) which parameters are ready."""

    def __init__(self, head, mod, samples, stream):
        self.head, self.mod = head, mod
        self.params = [p for p in mod.parameters() if p.requires_grad]
        self.static_in = tuple(t.detach().clone().requires_grad_(t.requires_grad) for t in samples)
        pool = torch.cuda.graph_pool_handle()
        self.fwd, self.bwd = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.fwd, pool=pool, stream=stream):
            outs = mod(*self.static_in)
        self.static_out = tuple(outs)
        self.static_gout = tuple(torch.empty_like(o) if o.requires_grad else None for o in outs)
        self.in_rg = [i for i, t in enumerate(self.static_in) if t.requires_grad]
        surface = [self.static_in[i] for i in self.in_rg] + self.params
        diff = [o for o in outs if o.requires_grad]
        grads = ()
        if diff and surface:
            with torch.cuda.graph(self.bwd, pool=pool, stream=stream):
                grads = torch.autograd.grad(diff, surface, [g for g in self.static_gout if g is not None], allow_unused=True)
        self.has_bwd = len(grads) > 0
        self.static_gin = [None] * len(self.static_in)
        for i, g in zip(self.in_rg, grads[:len(self.in_rg)]):
            self.static_gin[i] = g
        pg = grads[len(self.in_rg):]
        self.used = [(p, g) for p, g in zip(self.params, pg) if g is not None]      # parameters the chain really uses
        self.in_flight = False        # a forward whose backward has not run yet

    def _unalias(self):
        """Before ANY replay of this runner's graphs: a parameter whose `.grad` still IS its static buffer (handed over by the
        previous `_deliver` and neither consumed nor replaced since — `zero_grad(set_to_none=False)`, or a second micro-batch of
        gradient accumulation) gets a private copy.  The buffers live in the graphs' private pool, where the FORWARD graph's
        intermediates share their memory: the next forward replay already scribbles over them, the backward replay rewrites them,
        and `_deliver` would add a buffer to itself (found as garbage gradients by
        `test_chain_graphs_without_reducer_zero_in_place_and_accumulation`).  The usual loops (`set_to_none=True`, the bucketed
        reducer) never take this branch."""
        for p, g in self.used:
            if p.grad is not None and p.grad.data_ptr() == g.data_ptr():
                p.grad = g.clone()

    def _deliver(self):
        fresh = [(p, g) for p, g in self.used if p.grad is None]
        more = [(p, g) for p, g in self.used if p.grad is not None]
        for p, g in fresh:
            p.grad = g
        if more:
            torch._foreach_add_([p.grad for p, _ in more], [g for _, g in more])
        cb = getattr(self.head, 'on_param_grads', None)
        if cb is not None:
            cb([p for p, _ in self.used])


class _ChainGraphFn(torch.autograd.Function):

    @staticmethod
    def forward(ctx, runner, *inputs):
        for st, t in zip(runner.static_in, inputs):
            if st.data_ptr() != t.data_ptr():
                st.detach().copy_(t)
        runner._unalias()
        runner.fwd.replay()
        runner.in_flight = runner.has_bwd
        ctx.runner = runner
        return tuple(o.detach() for o in runner.static_out)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, *gouts):
        r = ctx.runner
        if not r.has_bwd:
            return (None,) * (1 + len(r.static_in))
        for st, g in zip(r.static_gout, gouts):
            if st is not None:
                if g is None:
                    st.zero_()
                elif st.data_ptr() != g.data_ptr():
                    st.copy_(g)
        r._unalias()
        r.bwd.replay()
        r.in_flight = False
        r._deliver()
        return (None,) + tuple(g.detach() if g is not None else None for g in r.static_gin)


@register_head
class KernelUpdateHead(nn.Module):

    def __init__(self, num_classes=80, num_ffn_fcs=2, num_heads=8, num_cls_fcs=1, num_mask_fcs=3,
                 feedforward_channels=2048, in_channels=256, out_channels=256, dropout=0.0, mask_thr=0.5,
                 act_cfg=dict(type='ReLU', inplace=True), ffn_act_cfg=dict(type='ReLU', inplace=True),
                 conv_kernel_size=3, feat_transform_cfg=None, hard_mask_thr=0.5, kernel_init=False, with_ffn=True,
                 mask_out_stride=4, relative_coors=False, relative_coors_off=False, feat_gather_stride=1,
                 mask_transform_stride=1, mask_upsample_stride=1, num_thing_classes=80, num_stuff_classes=53,
                 mask_assign_stride=4, ignore_label=255, thing_label_in_seg=0,
                 kernel_updator_cfg=dict(type='DynamicConv', in_channels=256, feat_channels=64, out_channels=256,
                                         input_feat_shape=1, act_cfg=dict(type='ReLU', inplace=True),
                                         norm_cfg=dict(type='LN')),
                 loss_rank=None,
                 loss_mask=dict(type='CrossEntropyLoss', use_mask=True, loss_weight=1.0),
                 loss_dice=dict(type='DiceLoss', loss_weight=3.0),
                 loss_cls=dict(type='FocalLoss', use_sigmoid=True, gamma=2.0, alpha=0.25, loss_weight=2.0),
                 **video_kwargs):
        super().__init__()
        self.num_classes = num_classes
        self.loss_cls = build_loss(loss_cls)
        self.loss_mask = build_loss(loss_mask)
        self.loss_dice = build_loss(loss_dice)
        self.loss_rank = build_loss(loss_rank) if loss_rank is not None else loss_rank
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.mask_thr = mask_thr
        self.fp16_enabled = False
        self.dropout = dropout
        self.num_heads = num_heads
        self.hard_mask_thr = hard_mask_thr
        self.kernel_init = kernel_init
        self.with_ffn = with_ffn
        self.mask_out_stride = mask_out_stride
        self.relative_coors = relative_coors
        self.relative_coors_off = relative_coors_off
        self.conv_kernel_size = conv_kernel_size
        self.feat_gather_stride = feat_gather_stride
        self.mask_transform_stride = mask_transform_stride
        self.mask_upsample_stride = mask_upsample_stride
        self.num_thing_classes = num_thing_classes
        self.num_stuff_classes = num_stuff_classes
        self.mask_assign_stride = mask_assign_stride
        self.ignore_label = ignore_label
        self.thing_label_in_seg = thing_label_in_seg
        self.feedforward_channels = feedforward_channels
        self.num_cls_fcs, self.num_mask_fcs = num_cls_fcs, num_mask_fcs

        if conv_kernel_size != 1:
            raise NotImplementedError('conv_kernel_size must be 1: every shipped config sets it '
                                      '(configs/det/_base_/models/knet_kitti_step_s3_r50_fpn.py:3)')
        if in_channels != out_channels:
            raise NotImplementedError('in_channels == out_channels (every shipped config)')
        if feat_gather_stride != 1 or mask_transform_stride != 1:
            raise NotImplementedError('feat_gather_stride / mask_transform_stride must be 1 (dead branches in shipped cfgs)')
        if max(num_cls_fcs, num_mask_fcs) > 4:
            raise NotImplementedError('at most 4 cls/mask fcs')

        E = in_channels * conv_kernel_size ** 2
        self.attention = _MHAParams(E, num_heads, dropout)
        self.attention_norm = nn.LayerNorm(E)
        self.kernel_update_conv = build_transformer_layer(kernel_updator_cfg)
        if feat_transform_cfg is not None:
            kernel_size = feat_transform_cfg.pop('kernel_size', 1)      # mutates the cfg dict like the reference (:108)
            if feat_transform_cfg.get('act_cfg', None) is not None or feat_transform_cfg.get('norm_cfg', None) is not None:
                raise NotImplementedError('feat_transform must be a bare conv (act_cfg=None, no norm; every shipped config)')
            self.feat_transform = _ConvParams(in_channels, kernel_size)
        else:
            self.feat_transform = None
        if self.with_ffn:
            self.ffn = _FFNParams(in_channels, feedforward_channels, num_ffn_fcs, dropout)
            self.ffn_norm = nn.LayerNorm(in_channels)
        self.cls_fcs = nn.ModuleList()
        for _ in range(num_cls_fcs):
            self.cls_fcs.append(nn.Linear(in_channels, in_channels, bias=False))
            self.cls_fcs.append(nn.LayerNorm(in_channels))
            self.cls_fcs.append(nn.ReLU(inplace=True))
        if self.loss_cls.use_sigmoid:
            self.fc_cls = nn.Linear(in_channels, self.num_classes)
        else:
            self.fc_cls = nn.Linear(in_channels, self.num_classes + 1)
        self.mask_fcs = nn.ModuleList()
        for _ in range(num_mask_fcs):
            self.mask_fcs.append(nn.Linear(in_channels, in_channels, bias=False))
            self.mask_fcs.append(nn.LayerNorm(in_channels))
            self.mask_fcs.append(nn.ReLU(inplace=True))
        self.fc_mask = nn.Linear(in_channels, out_channels)
        self._init_video(num_ffn_fcs=num_ffn_fcs, **video_kwargs)
        self._chain_graphs = None      # see enable_chain_graphs()
        self.vkn_flags = 0             # extra VKN_FLAG_* bits of every C call of this stage (e.g. ops.FLAG_CHAIN_PERSISTENT: A/B, tests)
        self._pack = None
        self._pack_sig = None

    def _init_video(self, **kw):
        extra = {k: v for k, v in kw.items() if k != 'num_ffn_fcs'}
        if extra:
            raise TypeError(f'unexpected keyword arguments {sorted(extra)}')
        self.previous = None
        self.previous_type = None

    # ---- reference: knet/det/kernel_update_head.py:151-168
    def init_weights(self):
        for p in self.parameters():
            if p.dim() > 1:
                nn.init.xavier_uniform_(p)
        if self.loss_cls.use_sigmoid:
            nn.init.constant_(self.fc_cls.bias, float(-math.log((1 - 0.01) / 0.01)))   # bias_init_with_prob(0.01)
        if self.kernel_init:
            nn.init.normal_(self.fc_mask.weight, mean=0, std=0.01)

    # ---- C-ABI plumbing
    def stage_pack(self, device):
        named = dict(self.named_parameters())
        sig = ops.StagePack.signature(named, device)
        if self._pack is None or sig != self._pack_sig:
            self._pack = ops.StagePack(named, device)
            self._pack_sig = sig
        return self._pack

    def link_packs(self, device):
        """(link_pre, link_track, track_src) `ops.link_pack`s of the previous-frame blocks (None for the plain heads)."""
        return None, None, 0

    def invalidate_pack(self):
        """Drop the cached C-ABI weight pack (pre-split bf16x3 images, composite weights).  The cache is keyed on every
        parameter's (data_ptr, _version); in-place writes through `param.data` (EMA hooks, some optimizers) bump neither — call
        this after such writes.  `load_state_dict` and `train()` / `eval()` transitions call it automatically."""
        self._pack = None
        self._pack_sig = None
        self._link_cache = None

    def train(self, mode=True):
        if mode != self.training:
            self.invalidate_pack()
        return super().train(mode)

    def _load_from_state_dict(self, *args, **kwargs):
        self.invalidate_pack()
        return super()._load_from_state_dict(*args, **kwargs)

    def make_dims(self, B, N, H, W):
        return ops.make_dims(B, N, self.in_channels, H, W, self.num_heads, self.feedforward_channels,
                             self.fc_cls.out_features, self.num_cls_fcs, self.num_mask_fcs, self.hard_mask_thr,
                             self.attention_norm.eps)

    def _check_inputs(self, x, proposal_feat, mask_preds, mask_shape):
        if not self.with_ffn:
            pass  # handled by the library (ffn pointers NULL)
        if x.dim() != 4 or mask_preds.dim() != 4:
            raise ValueError('x must be [B,C,H,W] and mask_preds [B,N,H,W]')

    @staticmethod
    def _gather_masks(x, mask_preds):
        """Reference :182-188: masks at another resolution than x are resized (bilinear, align_corners=False) before they are binarised
        for the gather.  No shipped config gets here; the resize is a torch op in front of the C call."""
        if tuple(mask_preds.shape[-2:]) != tuple(x.shape[-2:]):
            mask_preds = torch.nn.functional.interpolate(mask_preds, tuple(x.shape[-2:]), mode='bilinear', align_corners=False)
        return mask_preds

    @staticmethod
    def _to_mask_shape(x, new_mask_preds, mask_shape):
        """Reference :268-273: `mask_shape` (a caller-given output size) resizes the NEW mask logits when its height differs from x's."""
        if mask_shape is not None and mask_shape[0] != x.shape[-2]:
            new_mask_preds = torch.nn.functional.interpolate(new_mask_preds, tuple(mask_shape), mode='bilinear', align_corners=False)
        return new_mask_preds

    def _needs_grad(self, x, proposal_feat):
        """Autograd is wanted whenever grad mode is on and an input or any parameter requires grad — in train() AND eval() mode
        (a frozen / eval head inside a fine-tuned model still has to pass gradients to x and the kernels)."""
        return torch.is_grad_enabled() and (x.requires_grad or proposal_feat.requires_grad
                                            or any(p.requires_grad for p in self.parameters()))

    def _mha(self, mod, query, key=None, identity=None):
        """mmcv `MultiheadAttention.forward` (seq-first): identity + attn(q, k, v)[0], key = value, dropout 0."""
        key = query if key is None else key
        identity = query if identity is None else identity
        return identity + mod.attn(query, key, key, need_weights=False)[0]

    def _xfeat_autograd(self, x, mask_preds):
        """Differentiable gather: x [B,C,H,W], mask_preds [B,N,H,W] -> x_feat [B,N,C] with `feat_transform` folded
        (x_feat = xraw W^T + cnt b; reference :179-180, :190-195)."""
        if x.dtype not in (torch.float32, torch.float16, torch.bfloat16):
            raise TypeError(f'x: expected float32, float16 or bfloat16 features, got {x.dtype}')
        # (fp16 / bf16 x — BASELINE cfg2 "bf16", cfg5 "fp16": the forward passes are the half-storage kernels, bit-identical to the fp32
        #  ones on the rounded x; backward: dK on the widened x, dx rounded once to x's type — autograd.py, tests/test_gpu_xhalf.py)
        C = self.in_channels
        xraw, cnt = vag.mask_gather(x, mask_preds.detach(), self.hard_mask_thr)
        if self.feat_transform is None:
            return xraw
        if self.device_chain:
            B, N = xraw.shape[:2]
            folded = chain_train.linear(xraw.reshape(B * N, C), self.feat_transform.conv.weight.reshape(C, C)).view(B, N, C)
        else:
            folded = xraw @ self.feat_transform.conv.weight.reshape(C, C).t()
        return torch.addcmul(folded, cnt.unsqueeze(-1), self.feat_transform.conv.bias)      # (one launch: folded + cnt * bias)

    def _chain_autograd(self, x_feat, proposal_feat, previous_obj_feats=None):
        """The [B*N, C] chain as torch ops on this module's own parameters (reference :198-227; video :324-476):
        x_feat [B,N,C], proposal_feat [B,N,C,K,K] -> (cls_score | None, decode kernels [B,N,C], decode bias [B,N] | None,
        obj_feat [B,N,C,K,K], track | None).  `feat_transform` is folded into the decode kernels: Z = (mask_feat W) x + mask_feat . b."""
        B, N = proposal_feat.shape[:2]
        C, K = self.in_channels, self.conv_kernel_size
        pf = proposal_feat.reshape(B, N, C, -1).permute(0, 1, 3, 2)                                     # :198-199
        if previous_obj_feats is not None and getattr(self, 'previous_link', None) is not None:         # video :324-372
            pf = self._link_autograd(self._link_names('link'), x_feat, pf.reshape(B, N, C), previous_obj_feats,
                                     self.training and self.previous_detach_link).reshape(B, N, -1, C)
        obj_feat = _updator_torch(self.kernel_update_conv, x_feat, pf)                                 # :200
        obj_feat = obj_feat.reshape(B, N, -1).permute(1, 0, 2)                                          # :203-205
        obj_feat = self.attention_norm(self._mha(self.attention, obj_feat))                             # :206
        obj_feat = obj_feat.permute(1, 0, 2).reshape(B, N, -1, C)                                       # :208-211
        if self.with_ffn:
            obj_feat = self.ffn_norm(obj_feat + self.ffn.layers(obj_feat))                              # :214-215
        track = None
        if previous_obj_feats is not None and getattr(self, 'previous', None) is not None and self.previous_type is not None:   # video :394-476
            uf = {'ffn': None, 'update': x_feat, 'update_obj': obj_feat.reshape(B, N, C)}[self.previous_type]
            t = self._link_autograd(self._link_names('track'), uf, obj_feat.reshape(B, N, C), previous_obj_feats, False)
            track = t.reshape(B, N, C, K, K)
        cls_score = None
        if getattr(self, 'fc_cls', None) is not None:
            cls_feat = obj_feat.sum(-2)                                                                 # :217-221
            for layer in self.cls_fcs:
                cls_feat = layer(cls_feat)
            cls_score = self.fc_cls(cls_feat).view(B, N, -1)
        mask_feat = obj_feat
        for layer in self.mask_fcs:                                                                     # :223-227
            mask_feat = layer(mask_feat)
        mask_feat = self.fc_mask(mask_feat).reshape(B, N, C)
        if self.feat_transform is not None:                                                             # K (W x + b) = (K W) x + K.b
            kern = mask_feat @ self.feat_transform.conv.weight.reshape(C, C)
            kb = mask_feat @ self.feat_transform.conv.bias
        else:
            kern, kb = mask_feat, None
        return cls_score, kern, kb, obj_feat.permute(0, 1, 3, 2).reshape(B, N, C, K, K), track

    device_chain = True     # the training chain on the library's kernels (chain_train.py); False: torch autograd on library GEMMs

    def enable_device_chain(self, on=True):
        """Training: run the [B*N, C] chain — forward AND backward — on this library's kernels (`chain_train.chain_forward`, the
        default) or as torch autograd ops on the BLAS libraries' GEMMs (`_chain_autograd`: the A/B and the test oracle of the
        former).  Captured chain graphs are dropped (they hold the kernels of the previous setting)."""
        self.device_chain = bool(on)
        if getattr(self, '_chain_graphs', None) is not None:
            self._chain_graphs = {}
        return self

    def _chain_impl(self, x_feat, proposal_feat, previous_obj_feats=None):
        if self.device_chain and x_feat.is_cuda:
            # a shape the library's training kernels do not take (odd widths, wide LayerNorms, > 256 kernels per frame) runs the torch
            # autograd chain instead of raising in the middle of a step; asked once per (head, kernel count)
            key = int(proposal_feat.shape[1])
            ok = self._chain_supported.get(key) if hasattr(self, '_chain_supported') else None
            if ok is None:
                if not hasattr(self, '_chain_supported'):
                    self._chain_supported = {}
                ok = self._chain_supported[key] = chain_train.supported(self, key)
            if ok:
                return chain_train.chain_forward(self, x_feat, proposal_feat, previous_obj_feats)
        return self._chain_autograd(x_feat, proposal_feat, previous_obj_feats)

    def _forward_autograd(self, x, proposal_feat, mask_preds, previous_obj_feats=None):
        """Differentiable stage (training): the two x-streaming ops are the HIP kernels behind autograd Functions
        (video-k-net_amd/autograd.py: their backward passes are the same kernels with transposed operands), the [B*N, C] chain
        runs as torch ops on this module's own parameters.
        Counterpart of knet/det/kernel_update_head.py:170-277 (video: knet/video/kernel_update_head.py:281-541)."""
        if getattr(self, '_chain_graphs', None) is not None and x.is_cuda:
            # capture (first use of a shape) BEFORE this step's autograd graph touches the head's parameters: a live eager graph
            # through them at capture time takes the capture down (hipStreamEndCapture faults; measured, tools/micro/graph_capture_probe.py)
            xf_rg = x.requires_grad or (self.feat_transform is not None and any(p.requires_grad for p in self.feat_transform.parameters()))
            B, N = proposal_feat.shape[:2]
            self._chain_graph_for(x.new_empty((B, N, self.in_channels)).requires_grad_(xf_rg), proposal_feat, previous_obj_feats)
        x_feat = self._xfeat_autograd(x, mask_preds)
        cls_score, kern, kb, obj_feat, track = self._chain(x_feat, proposal_feat, previous_obj_feats)
        new_mask_preds = vag.mask_decode(x, kern, kb)                                                   # :247-260
        return cls_score, new_mask_preds, obj_feat, x_feat, track

    # ---- hipGraph capture of the chain (training): opt-in, `enable_chain_graphs()`
    def enable_chain_graphs(self, on=True):
        """Run the chain's forward and backward as captured hipGraphs (`torch.cuda.make_graphed_callables`): one graph pair per
        (shapes, requires-grad pattern, train / eval mode), captured at first use.  The values are those of the eager chain (same
        kernels, same order).  Parameters may be UPDATED in place (optimizers do) but not REPLACED: after `load_state_dict` with
        `assign=True`, `.to()` or re-initialisation call `enable_chain_graphs()` again (it drops the captured graphs).

        Streams: autograd binds a parameter's gradient-accumulation node to the stream it was created on, and the captured
        forward keeps those nodes alive.  When the training step runs on a NON-default stream (`with torch.cuda.stream(s):` around
        forward, backward and the optimizer), the capture happens on that same stream; on the default stream it needs a side stream.
        Either way the replays run on the current stream.

        Gradients: the parameters are constants of the captured graphs as far as autograd is concerned (`_ChainGraphRunner`): after
        a stage's backward replay their gradients are in `p.grad` (set, or added to an existing one) without per-parameter autograd
        nodes; `torch.autograd.grad(loss, parameters)` therefore does not see the chain's parameters.  A module may be in ONE
        forward at a time: a second forward before the first one's backward (recursive heads, the frame-sequential last stage of
        the previous_link heads) runs eagerly."""
        self._chain_graphs = {} if on else None
        return self

    on_param_grads = None     # callable(list of parameters): set by BucketedGradAllReducer; called when a captured backward has delivered

    def chain_graphs_new_step(self):
        """Forget forwards whose backward never ran (a training step that was abandoned): their graphs are usable again."""
        for runner, _ in (getattr(self, '_chain_graphs', None) or {}).values():
            runner.in_flight = False

    def _chain_graph_for(self, x_feat, proposal_feat, previous_obj_feats):
        """(captured graph pair, module, contiguous args) for these shapes / requires-grad flags / mode; captures at first use."""
        import gc
        args = [x_feat, proposal_feat] + ([previous_obj_feats] if previous_obj_feats is not None else [])
        args = [a.contiguous() for a in args]
        key = (self.training,) + tuple((tuple(a.shape), a.dtype, a.requires_grad) for a in args)
        graphs = self._chain_graphs
        if key not in graphs:
            gc.collect()                    # dead autograd graphs held by reference cycles count as "live" for the fault above
            mod = _ChainGraphModule(self, previous_obj_feats is not None)
            samples = tuple(torch.randn_like(a).requires_grad_(a.requires_grad) for a in args)
            cur = torch.cuda.current_stream(x_feat.device)
            own = cur != torch.cuda.default_stream(x_feat.device)
            # our own warm-up (lazy library initialisation, TunableOp's timing runs) instead of make_graphed_callables': that one
            # runs on an anonymous stream and its last outputs stay alive through the capture, so the parameters' accumulation
            # nodes would be the ones created THERE — bound to a stream nobody uses again — whatever stream the capture is on
            torch.cuda.synchronize()
            with torch.cuda.stream(cur if own else torch.cuda.Stream(device=x_feat.device)):
                surface = [t for t in samples if t.requires_grad] + [p for p in mod.parameters() if p.requires_grad]
                for _ in range(3):
                    outs = [o for o in mod(*samples) if o.requires_grad]
                    grads = torch.autograd.grad(outs, surface, [torch.empty_like(o) for o in outs], allow_unused=True)
                del outs, grads, surface
            torch.cuda.synchronize()
            gc.collect()
            # capture on the stream the step runs on when that is not the default stream (a capture cannot use the default one)
            graphs[key] = (_ChainGraphRunner(self, mod, samples, cur if own else None), mod)
        return graphs[key] + (args,)

    def _chain(self, x_feat, proposal_feat, previous_obj_feats=None):
        if getattr(self, '_chain_graphs', None) is None or not x_feat.is_cuda or not torch.is_grad_enabled():
            return self._chain_impl(x_feat, proposal_feat, previous_obj_feats)
        runner, mod, args = self._chain_graph_for(x_feat, proposal_feat, previous_obj_feats)
        if runner.in_flight or not any(a.requires_grad for a in args):
            # a second forward of the same module before its backward (recursive heads, frame-sequential previous_link stages) would
            # overwrite the activations the captured backward reads; inputs without gradients would cut the parameters off
            return self._chain_impl(x_feat, proposal_feat, previous_obj_feats)
        outs = iter(_ChainGraphFn.apply(runner, *args))
        return tuple(next(outs) if present else None for present in mod.pattern)

    def _link_names(self, which):
        raise NotImplementedError

    def _link_autograd(self, names, update_feature, cur, prev, detach_prev):
        """A link block in torch autograd (include/vkn.h: vkn_link_block_f32): cur, prev [B,N,C] -> [B,N,C]."""
        upd, att, norm, ffn, ffn_norm = (getattr(self, n) if n is not None else None for n in names)
        B, N, C = cur.shape
        prev = prev.reshape(B, N, C)
        if detach_prev:
            prev = prev.detach()
        if upd is not None:
            prev = _updator_torch(upd, update_feature, prev.reshape(B, N, 1, C)).reshape(B, N, C)
        q = cur.permute(1, 0, 2)
        t = norm(self._mha(att, q, prev.permute(1, 0, 2), q)).permute(1, 0, 2)
        return ffn_norm(t + ffn.layers(t))

    def _run(self, x, proposal_feat, mask_preds, previous_obj_feats=None, flags=0):
        B, N = proposal_feat.shape[:2]
        C, K = self.in_channels, self.conv_kernel_size
        H, W = x.shape[-2:]
        dims = self.make_dims(B, N, H, W)
        obj_in = proposal_feat.reshape(B, N, C)
        prev = previous_obj_feats.reshape(B, N, C) if previous_obj_feats is not None else None
        link_pre, link_track, track_src = self.link_packs(x.device) if prev is not None else (None, None, 0)
        cls, masks, obj, xfeat, track = ops.stage_forward(dims, self.stage_pack(x.device), x, obj_in, mask_preds, prev,
                                                          want_track=prev is not None and self.previous_type is not None,
                                                          flags=flags | getattr(self, 'vkn_flags', 0), link_pre=link_pre, link_track=link_track, track_src=track_src)
        obj = obj.reshape(B, N, C, K, K)
        if track is not None:
            track = track.reshape(B, N, C, K, K)
        return cls, masks, obj, xfeat, track

    def forward(self, x, proposal_feat, mask_preds, prev_cls_score=None, mask_shape=None, img_metas=None):
        """-> (cls_score [B,N,ncls], new_mask_preds [B,N,H,W], obj_feat [B,N,C,K,K])   reference :170-277"""
        self._check_inputs(x, proposal_feat, mask_preds, mask_shape)
        mask_preds = self._gather_masks(x, mask_preds)
        if self._needs_grad(x, proposal_feat):
            cls, masks, obj, _, _ = self._forward_autograd(x, proposal_feat, mask_preds)
        else:
            cls, masks, obj, _, _ = self._run(x, proposal_feat, mask_preds)
        return cls, self._to_mask_shape(x, masks, mask_shape), obj

    # ---- instance-only results: knet/det/kernel_update_head.py:443-481 (result formatting: K <= max_per_img masks per image)
    @staticmethod
    def _meta(cfg, key):
        return cfg[key] if isinstance(cfg, dict) else getattr(cfg, key)

    def rescale_masks(self, masks_per_img, img_meta):
        """sigmoid -> bilinear to batch_input_shape -> crop to img_shape -> bilinear to ori_shape (:443-458), on the masks' device
        (the K selected masks only; the panoptic path never materialises them — vkn_panoptic_joint_f32)."""
        import torch.nn.functional as F
        h, w = img_meta['img_shape'][:2]
        m = F.interpolate(masks_per_img.unsqueeze(0).sigmoid(), size=tuple(img_meta['batch_input_shape']), mode='bilinear',
                          align_corners=False)
        m = m[:, :, :h, :w]
        return F.interpolate(m, size=tuple(img_meta['ori_shape'][:2]), mode='bilinear', align_corners=False).squeeze(0)

    def get_seg_masks(self, masks_per_img, labels_per_img, scores_per_img, test_cfg, img_meta):
        """-> (bbox_result, segm_result) of mmdet's instance-segmentation format (:460-466)."""
        seg_masks = self.rescale_masks(masks_per_img, img_meta) > self._meta(test_cfg, 'mask_thr')
        return self.segm2result(seg_masks, labels_per_img, scores_per_img)

    def segm2result(self, mask_preds, det_labels, cls_scores):
        """One D2H copy of the K boolean masks; per-class lists exactly as the reference builds them (:468-481)."""
        import numpy as np
        num_classes = self.num_classes
        segm_result = [[] for _ in range(num_classes)]
        mask_preds = mask_preds.cpu().numpy()
        det_labels = det_labels.cpu().numpy()
        cls_scores = cls_scores.cpu().numpy()
        num_ins = mask_preds.shape[0]
        bboxes = np.zeros((num_ins, 5), dtype=np.float32)       # fake boxes: only the score column is filled
        bboxes[:, -1] = cls_scores
        bbox_result = [bboxes[det_labels == i, :] for i in range(num_classes)]
        for idx in range(num_ins):
            segm_result[det_labels[idx]].append(mask_preds[idx])
        return bbox_result, segm_result

    # ---- training: knet/det/kernel_update_head.py:279-441
    def loss(self, object_feats, cls_score, mask_pred, labels, label_weights, mask_targets, mask_weights, imgs_whwh=None,
             reduction_override=None, **kwargs):
        """Losses of one stage (reference :279-330).  Positives are the rows with a real class label.  When `labels` is the tensor
        `get_targets` just built, their flat indices are already known as a device tensor of STATIC length (matched predictions +
        present stuff rows): rows are then taken by index — no boolean-mask indexing, i.e. no `nonzero` and no device -> host
        synchronisation anywhere in the loss.  Foreign `labels` take the boolean path (same values)."""
        losses = dict()
        bg_class_ind = self.num_classes
        stash = getattr(self, '_targets_stash', None)
        pos_index = stash[1] if (stash is not None and stash[0] is labels) else None
        pos_inds = (labels >= 0) & (labels < bg_class_ind)
        num_pos = pos_inds.sum().float()
        avg_factor = reduce_mean(num_pos).clamp_(min=1.0)
        num_preds = mask_pred.shape[0] * mask_pred.shape[1]
        assert cls_score is None or (mask_pred.shape[0] == cls_score.shape[0] and mask_pred.shape[1] == cls_score.shape[1])
        take = (lambda t: t.index_select(0, pos_index)) if pos_index is not None else (lambda t: t[pos_inds])
        if cls_score is not None and cls_score.numel() > 0:
            flat_cls = cls_score.view(num_preds, -1)
            losses['loss_cls'] = self.loss_cls(flat_cls, labels, label_weights, avg_factor=avg_factor, reduction_override=reduction_override)
            losses['pos_acc'] = accuracy(take(flat_cls), take(labels))
        if mask_pred is not None:
            H, W = mask_pred.shape[-2:]
            any_pos = (pos_index.numel() > 0) if pos_index is not None else bool(pos_inds.any())
            if any_pos and pos_index is not None and self._fused_mask_losses_ok(mask_pred, reduction_override):
                # the three mask losses in two HIP passes forward and one backward (csrc/vkn_loss.hip) instead of ~60 element-wise /
                # reduction launches over [K, H, W] and [B, Ns, H, W] tensors; same values (tests/test_gpu_train.py)
                lm, ld, lr = vag.mask_losses(mask_pred, mask_targets, pos_index, self.loss_mask.loss_weight, self.loss_dice.loss_weight,
                                             self.loss_dice.eps, self.loss_rank.loss_weight if self.loss_rank is not None else None)
                losses['loss_mask'], losses['loss_dice'] = lm, ld
                if self.loss_rank is not None:
                    losses['loss_rank'] = lr
            elif any_pos:
                pos_mask_pred = take(mask_pred.reshape(num_preds, H, W))
                pos_mask_targets = take(mask_targets)
                losses['loss_mask'] = self.loss_mask(pos_mask_pred, pos_mask_targets)
                losses['loss_dice'] = self.loss_dice(pos_mask_pred, pos_mask_targets)
                if self.loss_rank is not None:
                    # reference :311-322 paints, per image, the positive kernels' target masks into a label map in ASCENDING kernel
                    # order (`for j in curr_rank: rank_target[i][mask_ij] = j`), i.e. every pixel ends up with the LARGEST positive
                    # kernel index whose target covers it.  Same map without the per-instance host loop (hundreds of tiny
                    # nonzero / index_put launches and a host sync each per step): a masked max over the kernel axis.
                    batch_size = mask_pred.size(0)
                    n = mask_targets.shape[0] // batch_size
                    covered = mask_targets.view(batch_size, n, H, W).bool() & pos_inds.view(batch_size, n, 1, 1)
                    idx = torch.arange(n, device=mask_targets.device, dtype=torch.int16).view(1, n, 1, 1)
                    top = torch.where(covered, idx, idx.new_full((), -1)).amax(dim=1).long()
                    rank_target = torch.where(top >= 0, top, top.new_full((), self.ignore_label))
                    losses['loss_rank'] = self.loss_rank(mask_pred, rank_target, ignore_index=self.ignore_label)
            else:
                losses['loss_mask'] = mask_pred.sum() * 0
                losses['loss_dice'] = mask_pred.sum() * 0
                if self.loss_rank is not None:
                    losses['loss_rank'] = mask_pred.sum() * 0
        return losses

    fused_mask_losses = True     # False: the torch op sequence (A/B; also taken whenever a loss object is not the shipped one)

    def _fused_mask_losses_ok(self, mask_pred, reduction_override):
        from . import losses as L
        from .train_tail import _is_sigmoid_dice
        lm, ld, lr = self.loss_mask, self.loss_dice, self.loss_rank
        return (self.fused_mask_losses and reduction_override is None and mask_pred.is_cuda and mask_pred.dtype == torch.float32
                and (mask_pred.shape[-1] * mask_pred.shape[-2]) % 4 == 0 and mask_pred.dim() == 4
                and type(lm) is L.CrossEntropyLoss and lm.use_sigmoid and lm.reduction == 'mean' and lm.class_weight is None
                and _is_sigmoid_dice(ld)        # (ours or mmdet's own class: checked by value, train_tail.py)
                and (lr is None or (type(lr) is L.CrossEntropyLoss and not lr.use_sigmoid and not lr.use_mask and lr.reduction == 'mean'
                                    and lr.class_weight is None)))

    def _get_target_single(self, pos_inds, neg_inds, pos_mask, neg_mask, pos_gt_mask, pos_gt_labels, gt_sem_seg, gt_sem_cls, cfg):
        """The reference's per-image entry point (:332-394) for callers that hold its argument list: the batch builder on one image
        (the matched rows and their ground truth are all it needs; the mask copies only give the sizes)."""
        from types import SimpleNamespace
        one = SimpleNamespace(pos_inds=pos_inds, pos_gt_masks=pos_gt_mask, pos_gt_labels=pos_gt_labels, num_pos=int(pos_mask.size(0)),
                              num_neg=int(neg_mask.size(0)), device=pos_mask.device, mask_dtype=pos_mask.dtype,
                              mask_shape=tuple(pos_mask.shape[1:]))
        sem = gt_sem_seg is not None and gt_sem_cls is not None
        out = self._batch_targets([one], cfg, [gt_sem_seg] if sem else None, [gt_sem_cls] if sem else None)
        self._targets_stash = None
        return out

    def get_targets(self, sampling_results, gt_mask, gt_labels, rcnn_train_cfg, concat=True, gt_sem_seg=None, gt_sem_cls=None):
        """(labels, label_weights, mask_targets, mask_weights) of a batch (reference :396-441).  `concat=True` (every caller in the
        reference): built for the WHOLE batch at once by `_batch_targets`; `concat=False`: per-image lists (the same builder, one
        image at a time)."""
        n = len(sampling_results)
        self._targets_stash = None
        if concat and n > 0:
            return self._batch_targets(sampling_results, rcnn_train_cfg, gt_sem_seg, gt_sem_cls)
        sem = gt_sem_seg is not None and gt_sem_cls is not None
        out = [self._batch_targets([r], rcnn_train_cfg, [gt_sem_seg[i]] if sem else None, [gt_sem_cls[i]] if sem else None)
               for i, r in enumerate(sampling_results)]
        self._targets_stash = None
        return tuple(list(t) for t in zip(*out)) if out else ([], [], [], [])

    def _batch_targets(self, sampling_results, cfg, gt_sem_seg, gt_sem_cls):
        """The targets of the images of a batch (reference :332-441), written directly in the batch layout
        [B, N + S] (N predictions, then the S stuff rows of that image) with a handful of scatters instead of ~25 small ops per
        image — and with the flat indices of the positive rows as a by-product (`loss` takes rows by them).  Same values:
          labels        num_classes, matched predictions <- their gt label, present stuff rows <- their class
          label_weights matched / unmatched predictions 1 (matched: pos_weight if > 0) over the thing columns only when stuff targets
                        exist (:388), stuff rows the identity over the stuff columns
          mask_targets  zeros, matched <- their gt mask, present stuff rows <- gt_sem_seg;  mask_weights 1 on exactly those rows."""
        r0 = sampling_results[0]
        B = len(sampling_results)
        dev, dt = r0.device, r0.mask_dtype
        H, W = r0.mask_shape[-2:]
        N = r0.num_pos + r0.num_neg
        with_sem = gt_sem_seg is not None and gt_sem_cls is not None and all(g is not None for g in gt_sem_seg) \
            and all(g is not None for g in gt_sem_cls)
        S, T = (self.num_stuff_classes, self.num_thing_classes) if with_sem else (0, 0)
        Ns, ncls = N + S, self.num_classes
        pw = cfg['pos_weight'] if isinstance(cfg, dict) else cfg.pos_weight
        pw = 1.0 if pw <= 0 else float(pw)
        pos_rows, pos_lab, pos_msk, sem_rows, sem_lab, sem_msk = [], [], [], [], [], []
        for i, r in enumerate(sampling_results):
            if r.num_pos + r.num_neg != N:
                raise ValueError('all images of a batch must carry the same number of predictions')
            pos_rows.append(r.pos_inds + i * Ns)
            pos_lab.append(r.pos_gt_labels)
            pos_msk.append(r.pos_gt_masks)
            if with_sem and gt_sem_cls[i].numel() > 0:
                cls_i = gt_sem_cls[i].to(dev).long()
                sem_rows.append(cls_i - T + (i * Ns + N))
                sem_lab.append(cls_i)
                sem_msk.append(gt_sem_seg[i])
        pos_rows = torch.cat(pos_rows)
        labels = torch.full((B * Ns,), ncls, dtype=torch.long, device=dev)
        label_weights = torch.zeros((B, Ns, ncls), dtype=dt, device=dev)
        mask_targets = torch.zeros((B * Ns, H, W), dtype=dt, device=dev)
        row_weight = torch.zeros((B * Ns,), dtype=dt, device=dev)
        label_weights[:, :N, :(T if with_sem else ncls)] = 1.0
        if pos_rows.numel() > 0:
            labels[pos_rows] = torch.cat(pos_lab)
            mask_targets[pos_rows] = torch.cat(pos_msk).to(dt)
            row_weight.index_fill_(0, pos_rows, 1.0)        # (x[idx] = python scalar would copy the scalar host -> device: a stall)
            if pw != 1.0:
                label_weights.view(B * Ns, ncls)[:, :(T if with_sem else ncls)][pos_rows] = pw
        if with_sem:
            label_weights[:, N:, T:] = torch.eye(S, dtype=dt, device=dev)
            if sem_rows:
                sem_rows = torch.cat(sem_rows)
                labels[sem_rows] = torch.cat(sem_lab)
                mask_targets[sem_rows] = torch.cat(sem_msk).to(dt)
                row_weight.index_fill_(0, sem_rows, 1.0)
                pos_rows = torch.sort(torch.cat([pos_rows, sem_rows]))[0]
        mask_weights = row_weight.view(-1, 1, 1).expand(-1, H, W)      # (a view: the reference fills a second [B*Ns, H, W] tensor)
        self._targets_stash = (labels, pos_rows)
        return labels, label_weights.view(B * Ns, ncls), mask_targets, mask_weights


@register_head
class VideoKernelUpdateHead(KernelUpdateHead):
    """knet/video/kernel_update_head.py:17-541.  `previous_type` (tracking embedding): 'ffn' (the r50 / VIP-Seg video configs,
    configs/det/video_knet_kitti_step/video_knet_s3_r50_*_link_ffn_joint_train.py:86-89), 'update', 'update_obj';
    `previous_link` (rewrites the stage's incoming kernels from the previous frame's): None, 'update_dynamic_cov'
    (configs/det/video_knet_kitti_step/video_knet_s3_swin{b,l}_*_joint_update.py:98-100, ..._update_conv_short_track_fc.py:95-97),
    'link_atten'.  Every block is one `vkn_link_block_f32`."""

    _TRACK_TYPES = (None, 'ffn', 'update', 'update_obj')
    _LINK_TYPES = (None, 'update_dynamic_cov', 'link_atten')

    def __init__(self, *args, previous=None, previous_type=None, previous_link=None, previous_x_feat=None,
                 previous_detach=False, previous_detach_link=False, previous_link_detach=False, **kwargs):
        self._video_cfg = dict(previous=previous, previous_type=previous_type, previous_link=previous_link,
                               previous_x_feat=previous_x_feat, previous_detach=previous_detach,
                               previous_detach_link=previous_detach_link, previous_link_detach=previous_link_detach)
        self._updator_cfg = kwargs.get('kernel_updator_cfg')
        super().__init__(*args, **kwargs)

    def _init_video(self, num_ffn_fcs=2, **kw):
        if kw:
            raise TypeError(f'unexpected keyword arguments {sorted(kw)}')
        for k, v in self._video_cfg.items():
            setattr(self, k, v)
        self._link_cache = None
        if self.previous is None:
            return
        if self.previous_type not in self._TRACK_TYPES or self.previous_link not in self._LINK_TYPES:
            raise ValueError(f'previous_type must be one of {self._TRACK_TYPES}, previous_link one of {self._LINK_TYPES}')
        E = self.in_channels * self.conv_kernel_size ** 2

        def block(sfx, updator):   # module names exactly as the reference builds them (:173-258)
            if updator:
                cfg = dict(self._updator_cfg) if self._updator_cfg is not None else dict(type='KernelUpdator')
                setattr(self, 'attention_previous_update' + sfx, build_transformer_layer(cfg))
            setattr(self, 'attention_previous' + sfx, _MHAParams(E, 8, 0.0))           # _num_head = 8, _dropout = 0. (:165-166)
            setattr(self, 'attention_previous_norm' + sfx, nn.LayerNorm(E))
            setattr(self, 'link_ffn' + sfx, _FFNParams(self.in_channels, self.feedforward_channels, num_ffn_fcs, self.dropout))
            setattr(self, 'link_ffn_norm' + sfx, nn.LayerNorm(self.in_channels))

        if self.previous_type == 'ffn':
            block('', False)
        elif self.previous_type in ('update', 'update_obj'):
            block('_track', True)
        if self.previous_link == 'update_dynamic_cov':
            block('_link', True)
        elif self.previous_link == 'link_atten':
            block('_link', False)

    def _link_names(self, which):
        """(updator, attention, norm, ffn, ffn_norm) attribute names of the tracking ('track') / previous_link ('link') block."""
        if which == 'link':
            sfx, upd = '_link', self.previous_link == 'update_dynamic_cov'
        else:
            sfx, upd = ('', False) if self.previous_type == 'ffn' else ('_track', True)
        return ('attention_previous_update' + sfx if upd else None, 'attention_previous' + sfx, 'attention_previous_norm' + sfx,
                'link_ffn' + sfx, 'link_ffn_norm' + sfx)

    def link_packs(self, device):
        if self.previous is None:
            return None, None, 0
        named = dict(self.named_parameters())
        sig = ops.StagePack.signature(named, device)
        if self._link_cache is None or self._link_cache[0] != sig:
            pre = ops.link_pack(named, device, *self._link_names('link')) if self.previous_link is not None else None
            trk, src = None, 0
            if self.previous_type in ('update', 'update_obj'):
                trk, src = ops.link_pack(named, device, *self._link_names('track')), 1 if self.previous_type == 'update' else 2
            self._link_cache = (sig, pre, trk, src)
        return self._link_cache[1:]

    def forward(self, x, proposal_feat, mask_preds, prev_cls_score=None, mask_shape=None, img_metas=None,
                previous_obj_feats=None, previous_mask_preds=None, previous_x_feats=None):
        """-> (cls_score, new_mask_preds, obj_feat, x_feat [B,N,C], object_feats_track [B,N,C,K,K] | None)  reference :281-541"""
        self._check_inputs(x, proposal_feat, mask_preds, mask_shape)
        if previous_obj_feats is not None and self.previous is None:
            previous_obj_feats = None      # no link modules were built (reference would fail on attribute access)
        if previous_obj_feats is not None and self.training and self.previous_detach:
            previous_obj_feats = previous_obj_feats.detach()                                               # :317-318
        mask_preds = self._gather_masks(x, mask_preds)
        if self._needs_grad(x, proposal_feat):
            cls, masks, obj, xfeat, track = self._forward_autograd(x, proposal_feat, mask_preds, previous_obj_feats)
        else:
            cls, masks, obj, xfeat, track = self._run(x, proposal_feat, mask_preds, previous_obj_feats)
        return cls, self._to_mask_shape(x, masks, mask_shape), obj, xfeat, track

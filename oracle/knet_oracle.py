"""ORACLE — TEST INFRASTRUCTURE ONLY.  Never imported by the product package.

CPU restatement (plain `torch`, functional, fp32 or fp64) of Video K-Net's kernel-update hot path,
written from the reference's behaviour, each function citing the reference file:line it follows
(paths relative to the upstream repo lxtGH/Video-K-Net).  Only `tests/`, `__graft_entry__.smoke()`
and `bench.py`'s `cpu_baseline` leg may import this module, and only as the checker / CPU baseline.

Pinning status: the reference ships NO tests and NO golden vectors.  This oracle
is pinned against outputs of the reference's own Python (imported unmodified in the build container
through `oracle/standins/`, see `oracle/gen_golden.py`) committed under `tests/golden/`.  The mmcv
wrapper semantics (`MultiheadAttention`, `FFN`, `ConvModule`) are not in the reference tree and mmcv is
not installable offline; they are restated from mmcv 1.3-1.7 — that part of the pin is conditional on
the restatement (flagged in DESIGN.md).

It uses the same ATen op sequence as the reference (1x1 `conv2d`, `sigmoid > thr`, `einsum`, linear,
`layer_norm`, softmax attention, per-image `conv2d`, bilinear `interpolate`) so that it also serves as the
"port" CPU baseline in bench.py.
"""
import math
from dataclasses import dataclass, field

import torch
import torch.nn.functional as F


@dataclass
class HeadCfg:
    """The subset of KernelUpdateHead / KernelIterHead ctor kwargs that changes forward arithmetic
    (knet/det/kernel_update_head.py:19-65, knet/det/kernel_iter_head.py:14-46)."""
    num_stages: int = 3
    in_channels: int = 256
    num_heads: int = 8
    num_classes: int = 19
    num_ffn_fcs: int = 2
    num_cls_fcs: int = 1
    num_mask_fcs: int = 1
    conv_kernel_size: int = 1
    hard_mask_thr: float = 0.5
    mask_upsample_stride: int = 2
    with_ffn: bool = True
    feat_transform: bool = True
    use_sigmoid_cls: bool = True
    feat_channels: int = 256      # KernelUpdator.feat_channels (== in_channels in every shipped cfg)
    previous_type: str = ''       # tracking embedding: 'ffn' (r50 / VIP-Seg video configs), 'update', 'update_obj'
                                  # (knet/video/kernel_update_head.py:173-214)
    previous_link: str = ''       # 'update_dynamic_cov' (swin "update" configs), 'link_atten': rewrites the incoming kernels (:216-258)
    ln_eps: float = 1e-5
    extra: dict = field(default_factory=dict)


def _ln(sd, pfx, x, eps):
    return F.layer_norm(x, (x.shape[-1],), sd[pfx + '.weight'], sd[pfx + '.bias'], eps)


def _linear(sd, pfx, x, bias=True):
    return F.linear(x, sd[pfx + '.weight'], sd[pfx + '.bias'] if bias else None)


def kernel_updator(sd, pfx, update_feature, input_feature, cfg: HeadCfg):
    """knet/kernel_updator.py:56-93 with the shipped flags gate_sigmoid=True, gate_norm_act=False,
    activate_out=False (:15-17).  update_feature [B,N,C]; input_feature [B,N,K*K,C] -> [B*N,K*K,C]."""
    Cin, Cf = cfg.in_channels, cfg.feat_channels
    u = update_feature.reshape(-1, Cin)                                   # :57
    n = u.size(0)
    params = _linear(sd, pfx + '.dynamic_layer', u)                       # :59
    param_in, param_out = params[:, :Cf], params[:, -Cf:]                 # :60-63
    feats = _linear(sd, pfx + '.input_layer', input_feature.reshape(n, -1, Cf))   # :65-66
    input_in, input_out = feats[..., :Cf], feats[..., -Cf:]               # :67-68
    gate = input_in * param_in.unsqueeze(-2)                              # :70
    input_gate = _ln(sd, pfx + '.input_norm_in', _linear(sd, pfx + '.input_gate', gate), cfg.ln_eps).sigmoid()   # :74,76-77
    update_gate = _ln(sd, pfx + '.norm_in', _linear(sd, pfx + '.update_gate', gate), cfg.ln_eps).sigmoid()       # :75,78
    param_out = _ln(sd, pfx + '.norm_out', param_out, cfg.ln_eps)         # :79
    input_out = _ln(sd, pfx + '.input_norm_out', input_out, cfg.ln_eps)   # :80
    feats = update_gate * param_out.unsqueeze(-2) + input_gate * input_out     # :87-88
    feats = _linear(sd, pfx + '.fc_layer', feats)                         # :90
    return F.relu(_ln(sd, pfx + '.fc_norm', feats, cfg.ln_eps))           # :91-92


def multihead_attention(sd, pfx, query, key, value, identity, num_heads):
    """mmcv `MultiheadAttention.forward` (identity + attn(q,k,v)[0], dropout 0, seq-first [L,B,E]) wrapping
    `torch.nn.MultiheadAttention` (packed in_proj, scaled dot-product softmax attention, out_proj).
    Call sites: knet/det/kernel_update_head.py:100-101,206; knet/video/kernel_update_head.py:174-178,404-411."""
    L, B, E = query.shape
    S = key.shape[0]
    hd = E // num_heads
    w, b = sd[pfx + '.attn.in_proj_weight'], sd[pfx + '.attn.in_proj_bias']
    q = F.linear(query, w[:E], b[:E])
    k = F.linear(key, w[E:2 * E], b[E:2 * E])
    v = F.linear(value, w[2 * E:], b[2 * E:])
    q = q.reshape(L, B * num_heads, hd).transpose(0, 1)      # [B*h, L, hd]
    k = k.reshape(S, B * num_heads, hd).transpose(0, 1)
    v = v.reshape(S, B * num_heads, hd).transpose(0, 1)
    attn = torch.softmax(torch.bmm(q * (1.0 / math.sqrt(hd)), k.transpose(1, 2)), dim=-1)
    out = torch.bmm(attn, v).transpose(0, 1).reshape(L, B, E)
    out = F.linear(out, sd[pfx + '.attn.out_proj.weight'], sd[pfx + '.attn.out_proj.bias'])
    return identity + out


def ffn(sd, pfx, x, num_fcs):
    """mmcv `FFN.forward` = x + layers(x), layers = Seq(Seq(Linear,ReLU,Drop)x(num_fcs-1), Linear, Drop), dropout 0.
    Call sites: knet/det/kernel_update_head.py:119-126,215."""
    h = x
    for i in range(num_fcs - 1):
        h = F.relu(_linear(sd, f'{pfx}.layers.{i}.0', h))
    h = _linear(sd, f'{pfx}.layers.{num_fcs - 1}', h)
    return x + h


def link_block(sd, pfx, sfx, with_updator, update_feature, cur, prev, cfg: HeadCfg):
    """The previous-frame blocks of `VideoKernelUpdateHead.forward` share one shape (knet/video/kernel_update_head.py):
         prev' = attention_previous_update{sfx}(update_feature, prev)          (only the "update" flavours: :332, :422, :451)
         t     = attention_previous_norm{sfx}(attention_previous{sfx}(query=cur, key=prev', value=prev', identity=cur))   (8 heads)
         out   = link_ffn_norm{sfx}(link_ffn{sfx}(t))
    sfx '' = previous_type 'ffn' (:394-415), '_track' = 'update' / 'update_obj' (:417-476), '_link' = previous_link (:324-372).
    cur, prev [B,N,C] (K = 1) -> [B,N,C]."""
    B, N, C = cur.shape
    if with_updator:
        prev = kernel_updator(sd, f'{pfx}.attention_previous_update{sfx}', update_feature, prev.reshape(B, N, 1, C), cfg).reshape(B, N, C)
    q, kv = cur.permute(1, 0, 2), prev.permute(1, 0, 2)
    t = multihead_attention(sd, f'{pfx}.attention_previous{sfx}', q, kv, kv, q, 8)                    # _num_head = 8 (:165, :219, :242)
    t = _ln(sd, f'{pfx}.attention_previous_norm{sfx}', t, cfg.ln_eps).permute(1, 0, 2)
    return _ln(sd, f'{pfx}.link_ffn_norm{sfx}', ffn(sd, f'{pfx}.link_ffn{sfx}', t, cfg.num_ffn_fcs), cfg.ln_eps)


def query_merge(sd, pfx, query, keys, pos=None, ln_eps=1e-5, heads=8):
    """Clip-level attention query merge of the VIS heads (knet_vis/tracker/kernel_frame_iter_head.py:142-160,
    knet_vis/tracker/kernel_update_head.py:244-263):
         t   = query_merge_norm(query_merge_attn(query=query, key=keys, value=keys, query_pos=pos, key_pos=pos per frame))   (8 heads)
         out = query_merge_ffn_norm(query_merge_ffn(t))
    mmcv's MultiheadAttention adds the positions to query / key only: value and the residual (identity) stay without them.
    query [B,N,C], keys [B,F*N,C] (frame-major), pos [N,C] | None -> [B,N,C].  `pfx` '' or 'mask_head.0' etc."""
    B, N, C = query.shape
    F_ = keys.shape[1] // N
    p = (pfx + '.') if pfx else ''
    q, k = query, keys
    if pos is not None:
        q = query + pos[None]
        k = keys + pos[None].repeat(1, F_, 1)
    t = multihead_attention(sd, p + 'query_merge_attn', q.permute(1, 0, 2), k.permute(1, 0, 2), keys.permute(1, 0, 2),
                            query.permute(1, 0, 2), heads).permute(1, 0, 2)   # (the reference builds it with 8 heads)
    t = _ln(sd, p + 'query_merge_norm', t, ln_eps)
    return _ln(sd, p + 'query_merge_ffn_norm', ffn(sd, p + 'query_merge_ffn', t, 2), ln_eps)


def binarize(mask_logits, thr):
    """knet/det/kernel_update_head.py:190-192: (sigmoid(z) > hard_mask_thr).float()."""
    return (mask_logits.sigmoid() > thr).to(mask_logits.dtype)


def mask_gather(x, m):
    """knet/det/kernel_update_head.py:195: einsum('bnhw,bchw->bnc')."""
    return torch.einsum('bnhw,bchw->bnc', m, x)


def mask_decode(x, mask_feat, K):
    """knet/det/kernel_update_head.py:247-260: per-image F.conv2d(x[i:i+1], mask_feat[i], padding=K//2)."""
    B = x.shape[0]
    outs = [F.conv2d(x[i:i + 1], mask_feat[i], padding=int(K // 2)) for i in range(B)]
    return torch.cat(outs, dim=0)


def update_head_stage(sd, pfx, x, proposal_feat, mask_preds, cfg: HeadCfg, previous_obj_feats=None, trace=None):
    """One `KernelUpdateHead.forward` (knet/det/kernel_update_head.py:170-277) / `VideoKernelUpdateHead.forward`
    with previous_type='ffn', previous_link=None (knet/video/kernel_update_head.py:281-541).
    Returns (cls_score [B,N,ncls], new_mask_preds [B,N,H,W], obj_feat [B,N,C,K,K], x_feat [B,N,C], track or None)."""
    B, N = proposal_feat.shape[:2]
    C, K = cfg.in_channels, cfg.conv_kernel_size
    if cfg.feat_transform:                                                 # :179-180 (ConvModule: conv only, bias, no norm/act)
        x = F.conv2d(x, sd[pfx + '.feat_transform.conv.weight'], sd[pfx + '.feat_transform.conv.bias'])
    H, W = x.shape[-2:]
    if mask_preds.shape[-2:] != (H, W):                                    # :183-186
        gather_mask = F.interpolate(mask_preds, (H, W), align_corners=False, mode='bilinear')
    else:
        gather_mask = mask_preds
    m = binarize(gather_mask, cfg.hard_mask_thr)                           # :190-192
    x_feat = mask_gather(x, m)                                             # :195
    pf = proposal_feat.reshape(B, N, C, -1).permute(0, 1, 3, 2)            # :198-200  [B,N,K*K,C]
    if previous_obj_feats is not None and cfg.previous_link:               # video :324-372: the incoming kernels are rewritten
        pf = link_block(sd, pfx, '_link', cfg.previous_link == 'update_dynamic_cov', x_feat, pf.reshape(B, N, C),
                        previous_obj_feats.reshape(B, N, C), cfg).reshape(B, N, 1, C)
    obj = kernel_updator(sd, pfx + '.kernel_update_conv', x_feat, pf, cfg)  # :201
    obj = obj.reshape(B, N, -1).permute(1, 0, 2)                           # :204-205  [N,B,K*K*C]
    obj_att = multihead_attention(sd, pfx + '.attention', obj, obj, obj, obj, cfg.num_heads)
    obj = _ln(sd, pfx + '.attention_norm', obj_att, cfg.ln_eps)            # :206
    obj = obj.permute(1, 0, 2).reshape(B, N, -1, C)                        # :208-211
    if cfg.with_ffn:                                                       # :214-215
        obj = _ln(sd, pfx + '.ffn_norm', ffn(sd, pfx + '.ffn', obj, cfg.num_ffn_fcs), cfg.ln_eps)
    track = None
    if previous_obj_feats is not None and cfg.previous_type == 'ffn':      # video :394-415
        prev = previous_obj_feats.reshape(B, N, C * K * K).permute(1, 0, 2)
        cur = obj.reshape(B, N, C * K * K).permute(1, 0, 2)
        t = multihead_attention(sd, pfx + '.attention_previous', cur, prev, prev, cur, 8)   # _num_head = 8, :165
        t = _ln(sd, pfx + '.attention_previous_norm', t, cfg.ln_eps)
        t = t.permute(1, 0, 2).reshape(B, N, -1, C)
        t = _ln(sd, pfx + '.link_ffn_norm', ffn(sd, pfx + '.link_ffn', t, cfg.num_ffn_fcs), cfg.ln_eps)
        track = t.permute(0, 1, 3, 2).reshape(B, N, C, K, K)               # :536-538
    elif previous_obj_feats is not None and cfg.previous_type in ('update', 'update_obj'):   # video :417-476
        uf = x_feat if cfg.previous_type == 'update' else obj.reshape(B, N, C)
        track = link_block(sd, pfx, '_track', True, uf, obj.reshape(B, N, C), previous_obj_feats.reshape(B, N, C), cfg)
        track = track.reshape(B, N, C, K, K)
    cls_feat = obj.sum(-2)                                                 # :217
    mask_feat = obj
    for i in range(cfg.num_cls_fcs):                                       # :220-221  Linear(no bias) + LN + ReLU
        cls_feat = F.relu(_ln(sd, f'{pfx}.cls_fcs.{3 * i + 1}', _linear(sd, f'{pfx}.cls_fcs.{3 * i}', cls_feat, bias=False), cfg.ln_eps))
    for i in range(cfg.num_mask_fcs):                                      # :222-223
        mask_feat = F.relu(_ln(sd, f'{pfx}.mask_fcs.{3 * i + 1}', _linear(sd, f'{pfx}.mask_fcs.{3 * i}', mask_feat, bias=False), cfg.ln_eps))
    cls_score = _linear(sd, pfx + '.fc_cls', cls_feat).view(B, N, -1)      # :225
    mask_feat = _linear(sd, pfx + '.fc_mask', mask_feat).permute(0, 1, 3, 2)   # :227
    mask_feat = mask_feat.reshape(B, N, C, K, K)                           # :244-246
    new_mask_preds = mask_decode(x, mask_feat, K).reshape(B, N, H, W)      # :247-260
    obj_out = obj.permute(0, 1, 3, 2).reshape(B, N, C, K, K)               # :275-277
    if trace is not None:
        trace.update(dict(x_t=x, bin_mask=m, x_feat=x_feat, cls_score=cls_score, mask_feat=mask_feat.reshape(B, N, C * K * K),
                          new_mask_preds=new_mask_preds, obj_feat=obj_out))
    return cls_score, new_mask_preds, obj_out, x_feat, track


def iter_head_mask_preds(sd, x, proposal_feats, mask_preds, cfg: HeadCfg, previous_obj_feats=None, traces=None,
                         prefix='mask_head'):
    """`KernelIterHead.simple_test_mask_preds` (knet/det/kernel_iter_head.py:285-311) and the video
    `simple_test_mask_preds_plus_previous` (knet/video/kernel_iter_head.py:529-564): S stages, bilinear
    x`mask_upsample_stride` on the last stage (`_mask_forward` :118-137), then cls activation (:307-310).
    `previous_obj_feats` is passed to the LAST stage only (video :544-546).
    Returns (object_feats, cls_score, mask_preds, scaled_mask_preds, object_feats_track or None)."""
    obj, cls, track = proposal_feats, None, None
    scaled = mask_preds
    for s in range(cfg.num_stages):
        prev = previous_obj_feats if s == cfg.num_stages - 1 else None
        tr = {} if traces is not None else None
        cls, mask_preds, obj, _xf, track = update_head_stage(sd, f'{prefix}.{s}', x, obj, mask_preds, cfg, prev, tr)
        if traces is not None:
            traces.append(tr)
        if cfg.mask_upsample_stride > 1 and s == cfg.num_stages - 1:
            scaled = F.interpolate(mask_preds, scale_factor=cfg.mask_upsample_stride, align_corners=False, mode='bilinear')
        else:
            scaled = mask_preds
    cls = cls.sigmoid() if cfg.use_sigmoid_cls else cls.softmax(-1)[..., :-1]
    return obj, cls, mask_preds, scaled, track


def stage_param_shapes(cfg: HeadCfg):
    """{state-dict key (without the `mask_head.{s}.` prefix): shape} of one stage — SURVEY.md §8(b)."""
    C, Cf, ncls = cfg.in_channels, cfg.feat_channels, cfg.num_classes
    E = C * cfg.conv_kernel_size ** 2
    sh = {}

    def lin(name, o, i, bias=True):
        sh[name + '.weight'] = (o, i)
        if bias:
            sh[name + '.bias'] = (o,)

    def ln(name, n):
        sh[name + '.weight'] = (n,)
        sh[name + '.bias'] = (n,)

    def mha(name):
        sh[name + '.attn.in_proj_weight'] = (3 * E, E)
        sh[name + '.attn.in_proj_bias'] = (3 * E,)
        lin(name + '.attn.out_proj', E, E)

    def ffn_(name):
        ff = cfg.extra.get('feedforward_channels', 2048)
        i = C
        for k in range(cfg.num_ffn_fcs - 1):
            lin(f'{name}.layers.{k}.0', ff, i)
            i = ff
        lin(f'{name}.layers.{cfg.num_ffn_fcs - 1}', C, i)

    mha('attention'); ln('attention_norm', E)
    ku = 'kernel_update_conv'
    lin(ku + '.dynamic_layer', 2 * Cf, C); lin(ku + '.input_layer', 2 * Cf, C)
    lin(ku + '.input_gate', Cf, C); lin(ku + '.update_gate', Cf, C)
    for n_ in ('norm_in', 'norm_out', 'input_norm_in', 'input_norm_out'):
        ln(f'{ku}.{n_}', Cf)
    lin(ku + '.fc_layer', C, Cf); ln(ku + '.fc_norm', C)
    if cfg.feat_transform:
        sh['feat_transform.conv.weight'] = (C, C, 1, 1)
        sh['feat_transform.conv.bias'] = (C,)
    if cfg.with_ffn:
        ffn_('ffn'); ln('ffn_norm', C)
    for i in range(cfg.num_cls_fcs):
        lin(f'cls_fcs.{3 * i}', C, C, bias=False); ln(f'cls_fcs.{3 * i + 1}', C)
    lin('fc_cls', ncls if cfg.use_sigmoid_cls else ncls + 1, C)
    for i in range(cfg.num_mask_fcs):
        lin(f'mask_fcs.{3 * i}', C, C, bias=False); ln(f'mask_fcs.{3 * i + 1}', C)
    lin('fc_mask', C, C)
    def updator(ku):
        lin(ku + '.dynamic_layer', 2 * Cf, C); lin(ku + '.input_layer', 2 * Cf, C)
        lin(ku + '.input_gate', Cf, C); lin(ku + '.update_gate', Cf, C)
        for n_ in ('norm_in', 'norm_out', 'input_norm_in', 'input_norm_out'):
            ln(f'{ku}.{n_}', Cf)
        lin(ku + '.fc_layer', C, Cf); ln(ku + '.fc_norm', C)

    def block(sfx, with_updator):
        if with_updator:
            updator('attention_previous_update' + sfx)
        mha('attention_previous' + sfx); ln('attention_previous_norm' + sfx, E)
        ffn_('link_ffn' + sfx); ln('link_ffn_norm' + sfx, C)

    if cfg.previous_type == 'ffn':
        block('', False)
    elif cfg.previous_type in ('update', 'update_obj'):
        block('_track', True)
    if cfg.previous_link:
        block('_link', cfg.previous_link == 'update_dynamic_cov')
    return sh


def head_param_shapes(cfg: HeadCfg, prefix='mask_head'):
    out = {}
    for s in range(cfg.num_stages):
        for k, v in stage_param_shapes(cfg).items():
            out[f'{prefix}.{s}.{k}'] = v
    return out


def kernel_init(init_w, loc_feats, semantic_feats=None, seg_w=None, seg_b=None, num_thing_classes=0, cat_stuff_mask=False,
                proposal_feats_with_obj=True, use_binary=True):
    """ConvKernelHead._decode_init_proposals after the loc / seg convs, eval mode        knet/det/kernel_head.py:204-263
    init_w [Np,C,1,1] = init_kernels.weight; seg_w [ncls,C,1,1], seg_b [ncls] = conv_seg.
    Returns (proposal_feats [B,N,C,1,1], x_feats, mask_preds [B,N,H,W], seg_preds | None)."""
    B = loc_feats.shape[0]
    Np, C = init_w.shape[0], init_w.shape[1]
    mask_preds = F.conv2d(loc_feats, init_w)                                                       # :222
    seg_preds = F.conv2d(semantic_feats, seg_w, seg_b) if semantic_feats is not None else None     # :231-234
    proposal_feats = init_w[None].expand(B, *init_w.shape)                                         # :234-236
    x_feats = semantic_feats + loc_feats if semantic_feats is not None else loc_feats              # :238-241
    if proposal_feats_with_obj:
        sig = mask_preds.sigmoid()                                                                 # :243-250
        nz = sig > 0.5
        w = nz.float() if use_binary else nz.float() * sig
        obj = torch.einsum('bnhw,bchw->bnc', w, x_feats)
        proposal_feats = proposal_feats + obj.view(B, Np, C, 1, 1)                                 # :252-254
    if cat_stuff_mask:                                                                             # :255-263 (eval)
        mask_preds = torch.cat([mask_preds, seg_preds[:, num_thing_classes:]], dim=1)
        stuff = seg_w[num_thing_classes:]
        proposal_feats = torch.cat([proposal_feats, stuff[None].expand(B, *stuff.shape)], dim=1)
    return proposal_feats, x_feats, mask_preds, seg_preds


def rescale_masks(masks_per_img, img_meta):
    """KernelUpdateHead.rescale_masks                                             knet/det/kernel_update_head.py:443-458"""
    h, w = img_meta['img_shape'][:2]
    m = F.interpolate(masks_per_img.unsqueeze(0).sigmoid(), size=tuple(img_meta['batch_input_shape']), mode='bilinear',
                      align_corners=False)
    m = m[:, :, :h, :w]
    return F.interpolate(m, size=tuple(img_meta['ori_shape'][:2]), mode='bilinear', align_corners=False).squeeze(0)


def panoptic_joint(cls_scores, mask_logits, num_proposals, num_thing_classes, max_per_img, instance_score_thr, overlap_thr,
                   img_meta, upsample_stride=1):
    """One image: the last-stage upsample of `_mask_forward` (knet/det/kernel_iter_head.py:122-130), then `get_panoptic`
    (:332-370) with merge_joint=True and `merge_stuff_thing_stuff_joint` (:467-524).
    cls_scores [N,ncls] (sigmoid applied), mask_logits [N,Hm,Wm].  Returns a dict with the reference's outputs
    (`panoptic_seg` int32, `segments_info`) and the intermediates the parity tests compare."""
    T = num_thing_classes
    scaled = mask_logits
    if upsample_stride > 1:
        scaled = F.interpolate(mask_logits[None], scale_factor=upsample_stride, align_corners=False, mode='bilinear')[0]
    thing_scores = cls_scores[:num_proposals][:, :T]
    thing_scores, topk_indices = thing_scores.flatten(0, 1).topk(max_per_img, sorted=True)            # :337-338
    mask_indices = topk_indices // T
    thing_labels = topk_indices % T
    thing_masks = rescale_masks(scaled[:num_proposals][mask_indices], img_meta)                       # :341-342
    stuff_scores = cls_scores[num_proposals:][:, T:].diag()                                           # :349-350
    stuff_scores, stuff_inds = torch.sort(stuff_scores, descending=True)
    stuff_masks = rescale_masks(scaled[num_proposals:][stuff_inds], img_meta)
    stuff_labels = stuff_inds + T                                                                     # :359
    total_masks = torch.cat([thing_masks, stuff_masks], dim=0)                                        # :480-482
    total_scores = torch.cat([thing_scores, stuff_scores], dim=0)
    total_labels = torch.cat([thing_labels, stuff_labels], dim=0)
    rows = torch.cat([mask_indices, stuff_inds + num_proposals], dim=0)
    cur_prob_masks = total_scores.view(-1, 1, 1) * total_masks
    cur_mask_ids = cur_prob_masks.argmax(0)                                                           # :486
    H, W = total_masks.shape[-2:]
    panoptic_seg = torch.zeros((H, W), dtype=torch.int32)
    segments_info = []
    K = total_masks.shape[0]
    area = torch.zeros(K, dtype=torch.int64)
    orig = torch.zeros(K, dtype=torch.int64)
    seg_of = torch.zeros(K, dtype=torch.int64)
    current_segment_id = 0
    for k in torch.argsort(-total_scores):                                                            # :489-522
        k = int(k)
        pred_class = int(total_labels[k])
        isthing = pred_class < T
        mask = cur_mask_ids == k
        area[k] = int(mask.sum())
        orig[k] = int((total_masks[k] >= 0.5).sum())
        if isthing and total_scores[k] < instance_score_thr:
            continue
        mask_area, original_area = int(area[k]), int(orig[k])
        if mask_area > 0 and original_area > 0:
            if mask_area / original_area < overlap_thr:
                continue
            current_segment_id += 1
            panoptic_seg[mask] = current_segment_id
            seg_of[k] = current_segment_id
            if isthing:
                segments_info.append(dict(id=current_segment_id, isthing=True, score=float(total_scores[k]),
                                          category_id=pred_class, instance_id=k))
            else:
                segments_info.append(dict(id=current_segment_id, isthing=False, category_id=pred_class - T + 1,
                                          area=mask_area))
    top2 = cur_prob_masks.topk(2, dim=0).values if K > 1 else None
    margin = (top2[0] - top2[1]) if top2 is not None else torch.full((H, W), float('inf'))
    return dict(panoptic_seg=panoptic_seg, segments_info=segments_info, cur_mask_ids=cur_mask_ids, total_scores=total_scores,
                total_labels=total_labels, rows=rows, area=area, orig=orig, seg_of=seg_of, margin=margin,
                total_masks=total_masks)


def assign_costs(mask_preds, cls_pred, gt_masks, gt_labels, cls_weight=2.0, dice_weight=4.0, mask_weight=1.0,
                 focal_alpha=0.25, focal_gamma=2.0, focal_eps=1e-12, dice_eps=1e-3, dice_pred_min=0.001, mask_pred_min=0.01):
    """Cost matrix of MaskHungarianAssigner.assign (knet/det/mask_hungarian_assigner.py:222-241) with FocalLossCost
    (mmdet 2.18, restated), DiceCost(pred_act=True) (:37-74) and MaskCost(pred_act=True) (:87-113).  -> [N, G] fp32.
    The knet_vis copies of the two cost classes take the plain sigmoid (knet_vis/det/mask_hungarian_assigner.py:69,100):
    dice_pred_min = mask_pred_min = 0."""
    p = cls_pred.sigmoid()                                                                            # FocalLossCost
    neg = -(1 - p + focal_eps).log() * (1 - focal_alpha) * p.pow(focal_gamma)
    pos = -(p + focal_eps).log() * focal_alpha * (1 - p).pow(focal_gamma)
    cls_cost = (pos[:, gt_labels] - neg[:, gt_labels]) * cls_weight
    mp = mask_preds.sigmoid().clamp(min=mask_pred_min, max=1.0)                                                # MaskCost :104-113
    H, W = gt_masks.shape[-2:]
    pos_c = torch.einsum('nhw,mhw->nm', mp, gt_masks)
    neg_c = torch.einsum('nhw,mhw->nm', 1 - mp, 1 - gt_masks)
    reg_cost = -(pos_c + neg_c) / (H * W) * mask_weight
    dp = mask_preds.sigmoid().clamp(min=dice_pred_min, max=1.0)                                               # DiceCost :48-74
    inp = dp.reshape(dp.shape[0], -1)
    tgt = gt_masks.reshape(gt_masks.shape[0], -1).float()
    a = torch.einsum('nh,mh->nm', inp, tgt)
    b = torch.sum(inp * inp, 1) + dice_eps
    c = torch.sum(tgt * tgt, 1) + dice_eps
    dice_cost = -((2 * a) / (b[:, None] + c[None, ...])) * dice_weight
    return cls_cost + reg_cost + dice_cost                                                            # :241


def hungarian_assign(cost, gt_labels):
    """Steps 3-4 of MaskHungarianAssigner.assign (:244-274): scipy LSAP on the host, 1-based gt indices, 0 = background."""
    from scipy.optimize import linear_sum_assignment
    rows, cols = linear_sum_assignment(cost.detach().cpu())
    n = cost.shape[0]
    gt_inds = torch.zeros(n, dtype=torch.long)
    labels = torch.full((n,), -1, dtype=torch.long)
    gt_inds[torch.from_numpy(rows)] = torch.from_numpy(cols) + 1
    labels[torch.from_numpy(rows)] = gt_labels[torch.from_numpy(cols)]
    return gt_inds, labels


def things_for_tracking(panoptic_seg, segments_info):
    """`get_things_id_for_tracking` (knet/video/knet_quansi_dense_embed_fc_joint_train.py:673-685) followed by
    `tensor_mask2box` (unitrack/utils/mask.py:41-46, 80-90) on the segment masks `panoptic_seg == id`:
    -> (instance ids, labels, boxes [n,4] = (xmin, ymin, xmax, ymax), scores) of the thing segments."""
    idxs, labels, boxes, scores = [], [], [], []
    seg = torch.as_tensor(panoptic_seg)
    for s in segments_info:
        if s['isthing']:
            m = (seg == s['id']).nonzero().float()          # rows of (y, x)
            if m.numel() > 0:                                # coords2bbox_all: (min col1, min col0, max col1, max col0)
                box = (m[:, 1].min().item(), m[:, 0].min().item(), m[:, 1].max().item(), m[:, 0].max().item())
            else:
                box = (-1, -1, 10, 10)
            idxs.append(s['instance_id']); labels.append(s['category_id']); boxes.append(box); scores.append(s['score'])
    return idxs, labels, boxes, scores

"""The [B*N, C] chain of a TRAINING step on the library's own kernels, forward and backward.

Until round 4 the chain of a training step was torch autograd on the BLAS libraries' GEMMs (`KernelUpdateHead._chain_autograd`); the
x-streaming ops on either side of it were already HIP in both directions (`autograd.py`).  Here every layer of the chain is an
`autograd.Function` over the C ABI:

  nn.Linear              forward   bf16x3 split-MFMA GEMM on the weight's tile images (`vkn_linear_f32`; the images are split from
                                   the CURRENT weights inside the call — weights change every step)
                         backward  dA = dY . W: the same GEMM kernel on the images of the TRANSPOSE (`vkn_split_weight_t_f32`);
                                   dW = dY^T . A, db = column sums of dY: `vkn_linear_dw_f32` (exact-fp32 MFMA, deterministic)
  nn.LayerNorm (+ ReLU / sigmoid behind it, + the residual added before it)       `vkn_layernorm_act_{fwd,bwd}_f32`
  the attention core of nn.MultiheadAttention                                     `vkn_attention_f32` / `vkn_attention_bwd_f32`

and `chain_forward` composes them exactly like `KernelUpdateHead._chain_autograd` (reference: knet/kernel_updator.py:56-93,
knet/det/kernel_update_head.py:198-227, the video links knet/video/kernel_update_head.py:324-476) — the torch chain stays as the A/B
and as the test oracle of this one (`tests/test_gpu_chain_train.py`).  What is left to torch inside the chain: reshapes, four
element-wise products / sums of the gated update, the `mask_feat . b_ft` row dot.  No BLAS call, no TunableOp.

Every Function allocates its scratch with `torch.empty` (never the cached `ops._workspace`): the chain is captured into hipGraphs
(`KernelUpdateHead.enable_chain_graphs`), and a captured pointer into a cache that is re-grown later would dangle.
"""
import ctypes

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib
from .ops import _ptr, _stream, check


def _f32c(t, name):
    if not t.is_cuda:
        raise _lib.VknLibraryError(f'{name}: expected a CUDA/HIP tensor — the MI355X path has no CPU fallback')
    if t.dtype != torch.float32:
        raise TypeError(f'{name}: expected float32, got {t.dtype}')
    t = t.contiguous()
    if t.data_ptr() % 16:
        t = t.clone(memory_format=torch.contiguous_format)
    return t


def _rows(t, name):
    """2-D fp32 with unit column stride (a column slice of a wider matrix is fine) -> (tensor, row stride)."""
    if t.dim() != 2 or t.dtype != torch.float32 or not t.is_cuda:
        raise TypeError(f'{name}: expected a 2-D float32 CUDA/HIP tensor')
    if t.stride(1) != 1 or t.data_ptr() % 4:
        t = t.contiguous()
    return t, t.stride(0)


def _rup(v, m):
    return (v + m - 1) // m * m


class WeightImages:
    """The bf16x3 tile images of a list of 2-D weights, BOTH orientations each, built from the current values by one launch per 32
    weights (`vkn_split_weights_batch_f32`) — the weights change every step, so this runs at the start of every chain forward and
    is part of the captured forward graph.  For a stored matrix W [R, Cc]:
      N = the images of W itself   ([Nout = R][K = Cc]):               y = a . W^T  (nn.Linear forward)    /  da = dy . W^T (wt)
      T = the images of W^T        ([Nout = Cc][K = roundup(R, 32)]):  da = dy . W  (nn.Linear backward)   /  y = a . W     (wt)"""

    def __init__(self, weights):
        self.map = {}
        todo = []
        for w in weights:
            key = (w.data_ptr(), tuple(w.shape))
            if key not in self.map and not any(k == key for k, _ in todo):
                todo.append((key, _f32c(w.detach(), 'weight')))
        if not todo:
            return
        dev = todo[0][1].device
        sizes = []
        for _, w in todo:
            R, Cc = w.shape
            if Cc % 32:
                raise ValueError(f'weight {tuple(w.shape)}: in features % 32 == 0')
            sizes.append((6 * _rup(R, 256) * Cc, 6 * _rup(Cc, 256) * _rup(R, 32)))
        total = sum(_rup(a, 256) + _rup(b, 256) for a, b in sizes)
        buf = torch.empty(total, dtype=torch.uint8, device=dev)
        items, off = [], 0
        for (key, w), (sn, st) in zip(todo, sizes):
            R, Cc = w.shape
            imgn, imgt = buf[off:off + sn], buf[off + _rup(sn, 256):off + _rup(sn, 256) + st]
            off += _rup(sn, 256) + _rup(st, 256)
            items.append(_lib.VknSplitItem(w.data_ptr(), imgn.data_ptr(), Cc, 1, R, Cc, Cc, 0))
            items.append(_lib.VknSplitItem(w.data_ptr(), imgt.data_ptr(), 1, Cc, Cc, _rup(R, 32), R, 0))
            self.map[key] = (imgn, imgt, w)            # (w: keeps a non-contiguous source's copy alive until the launch has run)
        L = _lib.lib()
        with torch.cuda.device(dev):
            for i in range(0, len(items), _lib.SPLIT_MAX_ITEMS):
                chunk = items[i:i + _lib.SPLIT_MAX_ITEMS]
                arr = (_lib.VknSplitItem * len(chunk))(*chunk)
                check(L.vkn_split_weights_batch_f32(arr, len(chunk), _stream()))

    def get(self, w):
        return self.map.get((w.data_ptr(), tuple(w.shape)), (None, None))[:2]


def _gemm(a, weight, images, bias, act, K, Nout):
    """act(a [M, K] . Wm^T + bias) with Wm [Nout, K] given by its tile images."""
    M = a.shape[0]
    out = torch.empty((M, Nout), dtype=torch.float32, device=a.device)
    ksplit, ws = 1, None
    if K >= 1024 and Nout <= 256:          # long contraction, few row tiles: split K over workgroups (fixed-order row epilogue)
        ksplit = 8
        ws = torch.empty(ksplit * M * Nout, dtype=torch.float32, device=a.device)
    check(_lib.lib().vkn_linear_f32(_ptr(a), _ptr(weight), _ptr(images), _ptr(bias), _ptr(out), M, K, Nout, int(act), ksplit,
                                    _ptr(ws), ws.numel() * 4 if ws is not None else 0, _stream()))
    return out


class LinearFn(torch.autograd.Function):
    """y = act(a . W^T + b) (wt=False, nn.Linear) or y = a . W + b (wt=True: the folded `feat_transform` weight, used as is).
    img_n / img_t: this weight's tile images from a `WeightImages` (None: built here)."""

    @staticmethod
    def forward(ctx, a, weight, bias, act, wt, img_n, img_t):
        a, weight = _f32c(a, 'a'), _f32c(weight, 'weight')
        if a.dim() != 2 or weight.dim() != 2:
            raise ValueError('LinearFn: a [M, K], weight [Nout, K] (or [K, Nout] with wt)')
        K = a.shape[1]
        Nout = weight.shape[1] if wt else weight.shape[0]
        if (weight.shape[0] if wt else weight.shape[1]) != K or K % 32:
            raise ValueError(f'LinearFn: shapes {tuple(a.shape)} x {tuple(weight.shape)} (in features % 32 == 0)')
        if wt and Nout % 32:
            raise ValueError('LinearFn(wt): out features % 32 == 0')
        bias = _f32c(bias, 'bias') if bias is not None else None
        if img_n is None or img_t is None:
            img_n, img_t = WeightImages([weight]).get(weight)
        with torch.cuda.device(a.device):
            y = _gemm(a, weight, img_t if wt else img_n, bias, act, K, Nout)
        ctx.act, ctx.wt, ctx.has_bias = act, wt, bias is not None
        ctx.save_for_backward(a, weight, y if act else None, img_n if wt else img_t)
        return y

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dy):
        a, weight, y, img_b = ctx.saved_tensors
        dy = _f32c(dy, 'dy')
        if ctx.act == 1:
            dy = dy * (y > 0)
        elif ctx.act:
            raise NotImplementedError
        M, K = a.shape
        Nout = dy.shape[1]
        L = _lib.lib()
        da = dw = db = None
        with torch.cuda.device(a.device):
            if ctx.needs_input_grad[0]:
                # wt: y = a . W -> da = dy . W^T, the images of W as stored;  else y = a . W^T -> da = dy . W, the images of W^T, whose
                # contraction length is the out-feature count rounded up to 32 (fc_cls: 19 classes) — dy is zero-padded to match
                g = dy if Nout % 32 == 0 else F.pad(dy, (0, 32 - Nout % 32))
                da = _gemm(g, weight, img_b, None, 0, g.shape[1], K)
            if ctx.needs_input_grad[1]:
                dw = torch.empty_like(weight)
                db = torch.empty(Nout, dtype=torch.float32, device=a.device) if (ctx.has_bias and ctx.needs_input_grad[2]) else None
                if ctx.wt:                                 # dW[k][n] = sum_m a[m][k] dy[m][n]
                    check(L.vkn_linear_dw_f32(_ptr(a), K, _ptr(dy), Nout, _ptr(dw), None, M, Nout, K, 0, _stream()))
                    if db is not None:
                        db = dy.sum(0)
                else:                                      # dW[n][k] = sum_m dy[m][n] a[m][k]
                    check(L.vkn_linear_dw_f32(_ptr(dy), Nout, _ptr(a), K, _ptr(dw), _ptr(db), M, K, Nout, 0, _stream()))
            elif ctx.has_bias and ctx.needs_input_grad[2]:
                db = dy.sum(0)
        return da, dw, db, None, None, None, None


class LayerNormActFn(torch.autograd.Function):
    """act(LayerNorm(x + resid)): act 0 none / 1 ReLU / 2 sigmoid.  x, resid: [M, C] with unit column stride (column slices ok)."""

    @staticmethod
    def forward(ctx, x, resid, gamma, beta, eps, act):
        x, ldx = _rows(x, 'x')
        M, C = x.shape
        ldr = 0
        if resid is not None:
            resid, ldr = _rows(resid, 'resid')
        out = torch.empty((M, C), dtype=torch.float32, device=x.device)
        stats = torch.empty((M, 2), dtype=torch.float32, device=x.device)
        with torch.cuda.device(x.device):
            check(_lib.lib().vkn_layernorm_act_fwd_f32(_ptr(x), ldx, _ptr(resid), ldr, _ptr(gamma), _ptr(beta), float(eps), int(act),
                                                       _ptr(out), C, _ptr(stats), M, C, _stream()))
        ctx.act, ctx.has_resid = int(act), resid is not None
        ctx.save_for_backward(x, resid, gamma, beta, stats)
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dy):
        x, resid, gamma, beta, stats = ctx.saved_tensors
        dy, lddy = _rows(dy, 'dy')
        M, C = x.shape
        L = _lib.lib()
        dx = torch.empty((M, C), dtype=torch.float32, device=x.device)
        want_p = gamma is not None and (ctx.needs_input_grad[2] or ctx.needs_input_grad[3])
        dg = torch.empty(C, dtype=torch.float32, device=x.device) if want_p else None
        dbt = torch.empty(C, dtype=torch.float32, device=x.device) if want_p else None
        with torch.cuda.device(x.device):
            check(L.vkn_layernorm_act_bwd_f32(_ptr(dy), lddy, _ptr(x), x.stride(0), _ptr(resid), resid.stride(0) if resid is not None else 0,
                                              _ptr(gamma), _ptr(beta), _ptr(stats), ctx.act, _ptr(dx), C, _ptr(dg), _ptr(dbt), M, C,
                                              _stream()))
        return (dx if ctx.needs_input_grad[0] else None, dx if (ctx.has_resid and ctx.needs_input_grad[1]) else None,
                dg if ctx.needs_input_grad[2] else None, dbt if ctx.needs_input_grad[3] else None, None, None)


class AttentionFn(torch.autograd.Function):
    """softmax(q k^T / sqrt(hd)) v per frame and head.  `kv` None: `q` is the packed in_proj output [B*N, 3C] (self-attention);
    else q [B*Nq, C] and kv [B*Nk, 2C] (cross-attention of the video links).  -> [B*Nq, C]"""

    @staticmethod
    def forward(ctx, q, kv, B, heads):
        q = _f32c(q, 'q')
        packed = kv is None
        C = q.shape[1] // 3 if packed else q.shape[1]
        src = q if packed else _f32c(kv, 'kv')
        if (q.shape[1] != 3 * C) if packed else (src.shape[1] != 2 * C):
            raise ValueError('AttentionFn: packed [M, 3C], or q [Mq, C] with kv [Mk, 2C]')
        Nq, Nk = q.shape[0] // B, src.shape[0] // B
        hd = C // heads
        out = torch.empty((q.shape[0], C), dtype=torch.float32, device=q.device)
        koff = C if packed else 0
        es = q.element_size()
        kp = ctypes.c_void_p(src.data_ptr() + koff * es)
        vp = ctypes.c_void_p(src.data_ptr() + (koff + C) * es)
        with torch.cuda.device(q.device):
            check(_lib.lib().vkn_attention_f32(_ptr(q), q.shape[1], kp, vp, src.shape[1], _ptr(out), C, B, Nq, Nk, heads, hd, _stream()))
        ctx.dims = (B, Nq, Nk, heads, hd, C, packed)
        ctx.save_for_backward(q, None if packed else src, out)
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, do):
        q, kv, out = ctx.saved_tensors
        B, Nq, Nk, heads, hd, C, packed = ctx.dims
        do = _f32c(do, 'do')
        src = q if packed else kv
        dq = torch.empty_like(q)                       # every (frame, head) workgroup writes its columns of every row: no zero fill
        dsrc = dq if packed else torch.empty_like(kv)
        koff = C if packed else 0
        es = 4
        with torch.cuda.device(q.device):
            check(_lib.lib().vkn_attention_bwd_f32(
                _ptr(q), q.shape[1], ctypes.c_void_p(src.data_ptr() + koff * es), ctypes.c_void_p(src.data_ptr() + (koff + C) * es),
                src.shape[1], _ptr(out), C, _ptr(do), C, _ptr(dq), dq.shape[1], ctypes.c_void_p(dsrc.data_ptr() + koff * es),
                ctypes.c_void_p(dsrc.data_ptr() + (koff + C) * es), dsrc.shape[1], B, Nq, Nk, heads, hd, _stream()))
        return dq, (None if packed else dsrc), None, None


# ---- functional forms
def linear(a, weight, bias=None, act=0, wt=False, images=None):
    img_n, img_t = images.get(weight) if images is not None else (None, None)
    return LinearFn.apply(a, weight, bias, act, wt, img_n, img_t)


def layernorm(x, norm: nn.LayerNorm, act=0, resid=None):
    return LayerNormActFn.apply(x, resid, norm.weight, norm.bias, norm.eps, act)


def attention(q, kv, B, heads):
    return AttentionFn.apply(q, kv, B, heads)


def _fc_stack(layers, t, imgs=None):
    """`cls_fcs` / `mask_fcs`: (Linear, LayerNorm, ReLU) triples (knet/det/kernel_update_head.py:124-152)."""
    mods = list(layers)
    i = 0
    while i < len(mods):
        m = mods[i]
        if isinstance(m, nn.Linear):
            t = linear(t, m.weight, m.bias, images=imgs)
            i += 1
        elif isinstance(m, nn.LayerNorm):
            relu = i + 1 < len(mods) and isinstance(mods[i + 1], nn.ReLU)
            t = layernorm(t, m, act=1 if relu else 0)
            i += 2 if relu else 1
        elif isinstance(m, nn.ReLU):
            t = torch.relu(t)
            i += 1
        else:
            raise NotImplementedError(type(m).__name__)
    return t


def kernel_updator(ku, update_feature, input_feature, imgs=None):
    """`KernelUpdator.forward` (knet/kernel_updator.py:56-93; K*K = 1): [M, C] x [M, C] -> [M, C]."""
    C = ku.feat_channels
    if getattr(ku, 'gate_norm_act', False) or getattr(ku, 'activate_out', False) or not getattr(ku, 'gate_sigmoid', True):
        raise NotImplementedError('KernelUpdator: gate_sigmoid=True, gate_norm_act=False, activate_out=False (every shipped config)')
    params = linear(update_feature, ku.dynamic_layer.weight, ku.dynamic_layer.bias, images=imgs)      # :58-63
    inputs = linear(input_feature, ku.input_layer.weight, ku.input_layer.bias, images=imgs)           # :65-68
    gate_feats = inputs[:, :C] * params[:, :C]                                                        # :70
    input_gate = layernorm(linear(gate_feats, ku.input_gate.weight, ku.input_gate.bias, images=imgs), ku.input_norm_in, act=2)   # :74, :79
    update_gate = layernorm(linear(gate_feats, ku.update_gate.weight, ku.update_gate.bias, images=imgs), ku.norm_in, act=2)      # :75, :80
    param_out = layernorm(params[:, C:], ku.norm_out)                                                 # :82
    input_out = layernorm(inputs[:, C:], ku.input_norm_out)                                           # :83
    features = update_gate * param_out + input_gate * input_out                                       # :89-90
    return layernorm(linear(features, ku.fc_layer.weight, ku.fc_layer.bias, images=imgs), ku.fc_norm, act=1)   # :92-93


def _ffn(ffn, t, imgs=None):
    """`FFN.layers` (mmcv): Seq(Seq(Linear, ReLU, Dropout) x (num_fcs - 1), Linear, Dropout), dropout 0."""
    for m in ffn.layers:
        if isinstance(m, nn.Sequential):
            t = linear(t, m[0].weight, m[0].bias, act=1, images=imgs)
        elif isinstance(m, nn.Linear):
            t = linear(t, m.weight, m.bias, images=imgs)
    return t


def link_block(head, names, update_feature, cur, prev, B, imgs=None):
    """A video link (`KernelUpdateHead._link_autograd`; include/vkn.h: vkn_link_block_f32): cur, prev [M, C] -> [M, C]."""
    upd, att, norm, ffn, ffn_norm = (getattr(head, n) if n is not None else None for n in names)
    C = cur.shape[1]
    if upd is not None:
        prev = kernel_updator(upd, update_feature, prev, imgs)
    w, b = att.attn.in_proj_weight, att.attn.in_proj_bias
    q = linear(cur, w[:C], b[:C], images=imgs)
    kv = linear(prev, w[C:], b[C:], images=imgs)
    o = linear(attention(q, kv, B, att.num_heads), att.attn.out_proj.weight, att.attn.out_proj.bias, images=imgs)
    t = layernorm(o, norm, resid=cur)
    return layernorm(_ffn(ffn, t, imgs), ffn_norm, resid=t)


# ---- the Linear weights a chain uses, in the order it uses them (one `WeightImages` per chain forward)
def _updator_weights(ku):
    return [ku.dynamic_layer.weight, ku.input_layer.weight, ku.input_gate.weight, ku.update_gate.weight, ku.fc_layer.weight]


def _ffn_weights(ffn):
    return [m[0].weight if isinstance(m, nn.Sequential) else m.weight for m in ffn.layers if isinstance(m, (nn.Sequential, nn.Linear))]


def _link_weights(head, names):
    upd, att, _, ffn, _ = (getattr(head, n) if n is not None else None for n in names)
    C = head.in_channels
    w = att.attn.in_proj_weight
    return (_updator_weights(upd) if upd is not None else []) + [w[:C], w[C:], att.attn.out_proj.weight] + _ffn_weights(ffn)


def chain_weights(head, has_prev):
    C = head.in_channels
    ws = []
    if has_prev and getattr(head, 'previous_link', None) is not None:
        ws += _link_weights(head, head._link_names('link'))
    ws += _updator_weights(head.kernel_update_conv) + [head.attention.attn.in_proj_weight, head.attention.attn.out_proj.weight]
    if head.with_ffn:
        ws += _ffn_weights(head.ffn)
    if has_prev and getattr(head, 'previous', None) is not None and head.previous_type is not None:
        ws += _link_weights(head, head._link_names('track'))
    ws += [m.weight for m in list(head.cls_fcs) + list(head.mask_fcs) if isinstance(m, nn.Linear)]
    if getattr(head, 'fc_cls', None) is not None:
        ws.append(head.fc_cls.weight)
    ws.append(head.fc_mask.weight)
    if head.feat_transform is not None:
        ws.append(head.feat_transform.conv.weight.reshape(C, C))
    return ws


def chain_forward(head, x_feat, proposal_feat, previous_obj_feats=None):
    """`KernelUpdateHead._chain_autograd` on the library's kernels: same arguments, same five results."""
    B, N = proposal_feat.shape[:2]
    C, K = head.in_channels, head.conv_kernel_size
    M = B * N
    xf = x_feat.reshape(M, C)
    pf = proposal_feat.reshape(M, C)                                                                  # K*K == 1
    prev = previous_obj_feats.reshape(M, C) if previous_obj_feats is not None else None
    imgs = WeightImages(chain_weights(head, prev is not None))       # every weight's tile images, both orientations: one launch
    if prev is not None and getattr(head, 'previous_link', None) is not None:                          # video :324-372
        p = prev.detach() if (head.training and head.previous_detach_link) else prev
        pf = link_block(head, head._link_names('link'), xf, pf, p, B, imgs)
    obj1 = kernel_updator(head.kernel_update_conv, xf, pf, imgs)                                      # :200
    mha = head.attention
    qkv = linear(obj1, mha.attn.in_proj_weight, mha.attn.in_proj_bias, images=imgs)
    ao = linear(attention(qkv, None, B, mha.num_heads), mha.attn.out_proj.weight, mha.attn.out_proj.bias, images=imgs)
    obj = layernorm(ao, head.attention_norm, resid=obj1)                                              # :206
    if head.with_ffn:
        obj = layernorm(_ffn(head.ffn, obj, imgs), head.ffn_norm, resid=obj)                          # :214-215
    track = None
    if prev is not None and getattr(head, 'previous', None) is not None and head.previous_type is not None:   # video :394-476
        uf = {'ffn': None, 'update': xf, 'update_obj': obj}[head.previous_type]
        track = link_block(head, head._link_names('track'), uf, obj, prev, B, imgs).reshape(B, N, C, K, K)
    cls_score = None
    if getattr(head, 'fc_cls', None) is not None:
        cls_score = linear(_fc_stack(head.cls_fcs, obj, imgs), head.fc_cls.weight, head.fc_cls.bias, images=imgs).view(B, N, -1)   # :217-221
    mask_feat = linear(_fc_stack(head.mask_fcs, obj, imgs), head.fc_mask.weight, head.fc_mask.bias, images=imgs)   # :223-227
    if head.feat_transform is not None:                                                               # K (W x + b) = (K W) x + K.b
        ft = head.feat_transform.conv
        kern = linear(mask_feat, ft.weight.reshape(C, C), wt=True, images=imgs)
        kb = (mask_feat * ft.bias).sum(-1).view(B, N)
    else:
        kern, kb = mask_feat, None
    return cls_score, kern.view(B, N, C), kb, obj.reshape(B, N, C, K, K), track

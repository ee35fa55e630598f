"""Python <-> C-ABI glue for the kernel-update head ops: torch is used only for device memory and streams.

Every function enqueues HIP kernels of libvkn.so on the CURRENT torch stream, on the tensors' device, and returns
torch tensors.  Inputs must be CUDA(=HIP) fp32 tensors; there is no CPU path here by design — the CPU restatement
lives in `oracle/` and is test infrastructure only.
"""
import ctypes
import struct
import threading

import torch

from . import _lib
from ._lib import VknDims, VknStageWeights, check

FLAG_REF_KERNELS = 1
FLAG_EXACT_GEMM = 2
FLAG_LOGITS_HANDOFF = 4
FLAG_BITS_HANDOFF = 16
FLAG_CHAIN_LAUNCHES = 256    # the [N x C] chain always as one launch per GEMM (default: by row count, include/vkn.h)
FLAG_CHAIN_PERSISTENT = 512  # ... always as the two persistent row-owner kernels (vkn_chain.hip)
FLAG_JOIN_EARLY = 32768     # head_forward: the side-stream link joins BEFORE the upsample (single-call latency; 1-3 % slower in throughput)
FLAG_SCALED_F16 = 16384     # head_forward: the up-scaled logits as fp16 (vkn_upsample_bilinear_f16out)
FLAG_CHAIN_BF16X3 = 65536   # ... the persistent kernels on the three-term bf16 split of rounds 2-4 (default since round 5: the two-term fp16 split, vkn_chain_h2.hip)
FLAG_INIT_SEPARATE = 131072   # vkn_kernel_init_f32: the round-5 form of pass 0 (separate decodes + add + logits gather) instead of the one-pass kernel (A/B)
FLAG_CHAIN_KSPLIT = 8192     # ... always as the few-row chain: column-spread GEMM phases, normalisation in the consumer (vkn_ksplit.hip)
FLAG_SERIAL_LINK = 32   # tracking link on the caller's stream instead of the library's side stream (A/B; same results)

_tls = threading.local()


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _req(t, name):
    if not torch.is_tensor(t) or not t.is_cuda:
        raise _lib.VknLibraryError(f'{name}: expected a CUDA/HIP tensor — the MI355X path has no CPU fallback')
    if t.dtype != torch.float32:
        raise TypeError(f'{name}: expected float32, got {t.dtype}')
    t = t.contiguous()
    if t.data_ptr() % 16:
        t = t.clone(memory_format=torch.contiguous_format)
    return t


_X_DTYPES = {torch.float32: 0, torch.float16: 1, torch.bfloat16: 2}
FLAG_X_F16, FLAG_X_BF16 = 64, 128


def _req_x(t, name='x'):
    """The feature map: fp32, or fp16 / bf16 STORAGE (VKN_X_*; the head still computes in fp32).  Returns (tensor, x_dtype code)."""
    if not torch.is_tensor(t) or not t.is_cuda:
        raise _lib.VknLibraryError(f'{name}: expected a CUDA/HIP tensor — the MI355X path has no CPU fallback')
    if t.dtype not in _X_DTYPES:
        raise TypeError(f'{name}: expected float32, float16 or bfloat16, got {t.dtype}')
    t = t.contiguous()
    if t.data_ptr() % 16:
        t = t.clone(memory_format=torch.contiguous_format)
    return t, _X_DTYPES[t.dtype]


def _workspace(nbytes, device, scratch=False):
    """Grow-only per-(thread, device, stream) scratch buffer handed to the C ABI (the library never allocates).  Keyed by the
    current stream so that calls enqueued on different streams never share scratch memory.
    The first 256 bytes are the HEADER of the stage / head / chain entry points (their status word, include/vkn.h: they carve it
    first and only ever OR into it).  Every other entry point uses its workspace from offset 0 — `scratch=True` hands those the
    buffer BEHIND the header, so that a gather / decode / kernel-init / assignment call between two head calls cannot leave its
    scratch data where `workspace_status` reads the status word (found by the soak: a spurious error after the kernel-init pass)."""
    cache = getattr(_tls, 'ws', None)
    if cache is None:
        cache = _tls.ws = {}
    key = (device, torch.cuda.current_stream(device).cuda_stream)
    buf = cache.get(key)
    need = int(nbytes) + 256
    if buf is None or buf.numel() < need:
        old = buf
        buf = torch.empty(max(need, 512), dtype=torch.uint8, device=device)
        if old is None:
            buf[:256].zero_()      # the workspace header: its first word is the status word the kernels OR into (include/vkn.h)
        else:
            buf[:256].copy_(old[:256])   # regrow: a sticky VKN_STATUS_RANGE nobody has read yet moves with the header (same stream: ordered)
        cache[key] = buf
        gen = getattr(_tls, 'ws_gen', None)
        if gen is None:
            gen = _tls.ws_gen = {}
        gen[key] = gen.get(key, 0) + 1   # `workspace_generation`: state kept in the buffer between calls (PHASE_A/B/C) dies with a regrow
    return buf[256:] if scratch else buf


def workspace_generation(device):
    """How often this (thread, device, stream)'s workspace has been (re)allocated.  The phased clip call (PHASE_A / B / C) keeps state
    in the workspace between its three calls: a caller compares the generation before B / C with the one after A."""
    gen = getattr(_tls, 'ws_gen', None) or {}
    return gen.get((device, torch.cuda.current_stream(device).cuda_stream), 0)


def workspace_status(device=None):
    """Synchronise the current stream and read-and-clear the status word of this (thread, device, stream)'s workspace: raises
    `VknError` (VKN_E_RANGE) when a mask gather of a stage / head call since the last check produced non-finite sums — some
    |x| >= 65504 or a non-finite x entered the f16 split (include/vkn.h: VKN_STATUS_RANGE).  Costs one stream synchronisation: call it
    when you would look at the results anyway (per video, per evaluation step), not per frame."""
    device = torch.device(device if device is not None else torch.cuda.current_device())
    if device.index is None:
        device = torch.device('cuda', torch.cuda.current_device())
    cache = getattr(_tls, 'ws', None) or {}
    buf = cache.get((device, torch.cuda.current_stream(device).cuda_stream))
    if buf is None:
        return
    with torch.cuda.device(device):
        check(_lib.lib().vkn_workspace_status(_ptr(buf), buf.numel(), _stream()))


_THR_CACHE = {}


def thr_logit(hard_mask_thr=0.5):
    """Smallest fp32 z with `torch.sigmoid(z) > hard_mask_thr` as ATen's CPU fp32 sigmoid evaluates it — the reference's
    `(mask_preds.sigmoid() > hard_mask_thr)` (knet/det/kernel_update_head.py:190-191) becomes `z >= thr_logit`.
    For 0.5 this is 8.940697e-08, NOT 0 (1/(1+exp(-z)) rounds to exactly 0.5 for smaller positive z).
    Found by bisection over the ordered fp32 bit patterns (sigmoid is monotone)."""
    key = float(hard_mask_thr)
    if key in _THR_CACHE:
        return _THR_CACHE[key]
    if not 0.0 < key < 1.0:
        raise ValueError('hard_mask_thr must be in (0, 1)')

    def f2o(f):  # float -> monotone integer key
        u = struct.unpack('<I', struct.pack('<f', f))[0]
        return u ^ 0xFFFFFFFF if u & 0x80000000 else u | 0x80000000

    def o2f(o):
        u = o & 0x7FFFFFFF if o & 0x80000000 else o ^ 0xFFFFFFFF
        return struct.unpack('<f', struct.pack('<I', u & 0xFFFFFFFF))[0]

    def on(o):
        # evaluate through a padded vector so the vectorised kernel (not only the scalar tail) is exercised
        t = torch.full((16,), o2f(o), dtype=torch.float32)
        return bool((t.sigmoid() > key)[0])

    lo, hi = f2o(-200.0), f2o(200.0)
    assert not on(lo) and on(hi)
    while hi - lo > 1:
        mid = (lo + hi) // 2
        if on(mid):
            hi = mid
        else:
            lo = mid
    _THR_CACHE[key] = o2f(hi)
    return _THR_CACHE[key]


def mask_gather(x, mask_logits, hard_mask_thr=0.5, flags=0):
    """(xraw [B,N,C], cnt [B,N]) = sum_p bit(mask)[b,n,p] * x[b,c,p] and the ON-pixel count.
    Replaces `einsum('bnhw,bchw->bnc', (mask.sigmoid() > thr).float(), x)` (knet/det/kernel_update_head.py:190-195)."""
    (x, xdt), m = _req_x(x), _req(mask_logits, 'mask_logits')
    flags |= (0, FLAG_X_F16, FLAG_X_BF16)[xdt]
    B, C = x.shape[0], x.shape[1]
    N = m.shape[1]
    P = x[0, 0].numel()
    if m.shape[0] != B or m[0, 0].numel() != P:
        raise ValueError('x and mask_logits disagree on batch or spatial size')
    L = _lib.lib()
    xraw = torch.empty((B, N, C), dtype=torch.float32, device=x.device)
    cnt = torch.empty((B, N), dtype=torch.float32, device=x.device)
    nb = L.vkn_gather_workspace_bytes(B, N, C, P)
    ws = _workspace(nb, x.device, scratch=True)
    with torch.cuda.device(x.device):
        check(L.vkn_mask_gather_f32(_ptr(x), _ptr(m), thr_logit(hard_mask_thr), _ptr(xraw), _ptr(cnt), B, N, C, P, _ptr(ws),
                                    ws.numel(), flags, _stream()))
    return xraw, cnt


def mask_gather_real(x, a):
    """(out [B,N,C], asum [B,N]) = sum_p a[b,n,p] * x[b,c,p] and sum_p a — the gather with a REAL-valued left operand
    (`einsum('bnhw,bchw->bnc', a, x)`): decode backward w.r.t. the kernels, soft gather weights, soft ground-truth masks."""
    x, a = _req(x, 'x'), _req(a, 'a')
    B, C = x.shape[0], x.shape[1]
    N = a.shape[1]
    P = x[0, 0].numel()
    if a.shape[0] != B or a[0, 0].numel() != P:
        raise ValueError('x and a disagree on batch or spatial size')
    L = _lib.lib()
    out = torch.empty((B, N, C), dtype=torch.float32, device=x.device)
    asum = torch.empty((B, N), dtype=torch.float32, device=x.device)
    ws = _workspace(L.vkn_gather_workspace_bytes(B, N, C, P), x.device, scratch=True)
    with torch.cuda.device(x.device):
        check(L.vkn_mask_gather_real_f32(_ptr(x), _ptr(a), _ptr(out), _ptr(asum), B, N, C, P, _ptr(ws), ws.numel(), _stream()))
    return out, asum


def mask_decode(x, kernels, bias=None, flags=0, out_scale=None):
    """out[b,n,h,w] = sum_c kernels[b,n,c] x[b,c,h,w] (+ bias[b,n]).
    Replaces the per-image `F.conv2d(x[i:i+1], mask_feat[i])`, K=1 (knet/det/kernel_update_head.py:247-260).
    `out_scale` (a DEVICE scalar tensor): every output times it, inside the kernel (the backward passes' power-of-two unscaling)."""
    (x, xdt), k = _req_x(x), _req(kernels.reshape(kernels.shape[0], kernels.shape[1], -1), 'kernels')
    flags |= (0, FLAG_X_F16, FLAG_X_BF16)[xdt]
    B, C, H, W = x.shape
    N = k.shape[1]
    if k.shape[0] != B or k.shape[2] != C:
        raise ValueError('kernels must be [B, N, C] (conv_kernel_size == 1)')
    bias = _req(bias, 'bias') if bias is not None else None
    L = _lib.lib()
    out = torch.empty((B, N, H, W), dtype=torch.float32, device=x.device)
    nb = L.vkn_decode_workspace_bytes(B, N, C)
    ws = _workspace(nb, x.device, scratch=True)
    with torch.cuda.device(x.device):
        if out_scale is not None:
            sc = _req(out_scale.reshape(1), 'out_scale')
            check(L.vkn_mask_decode_scaled_f32(_ptr(x), _ptr(k), _ptr(bias), _ptr(sc), _ptr(out), B, N, C, H * W, _ptr(ws), ws.numel(),
                                               flags, _stream()))
        else:
            check(L.vkn_mask_decode_f32(_ptr(x), _ptr(k), _ptr(bias), _ptr(out), B, N, C, H * W, _ptr(ws), ws.numel(), flags,
                                        _stream()))
    return out


def upsample_bilinear(masks, scale, out_f16=False):
    """`F.interpolate(masks, scale_factor=scale, mode='bilinear', align_corners=False)` (knet/det/kernel_iter_head.py:122-130).
    out_f16 (opt-in; scale 2 / 4): the result as fp16 — the same fp32 interpolation rounded once at the store, half the bytes."""
    m = _req(masks, 'masks')
    B, N, H, W = m.shape
    out = torch.empty((B, N, H * scale, W * scale), dtype=torch.float16 if out_f16 else torch.float32, device=m.device)
    with torch.cuda.device(m.device):
        if out_f16:
            check(_lib.lib().vkn_upsample_bilinear_f16out(_ptr(m), _ptr(out), B * N, H, W, int(scale), _stream()))
        else:
            check(_lib.lib().vkn_upsample_bilinear_f32(_ptr(m), _ptr(out), B * N, H, W, int(scale), _stream()))
    return out


def upsample_bilinear_bwd(grad_out, scale):
    """Adjoint of `upsample_bilinear`: grad_out [B, N, H*scale, W*scale] -> [B, N, H, W] (vkn_upsample_bilinear_bwd_f32)."""
    g = _req(grad_out, 'grad_out')
    B, N, OH, OW = g.shape
    H, W = OH // scale, OW // scale
    out = torch.empty((B, N, H, W), dtype=torch.float32, device=g.device)
    with torch.cuda.device(g.device):
        check(_lib.lib().vkn_upsample_bilinear_bwd_f32(_ptr(g), _ptr(out), B * N, H, W, int(scale), _stream()))
    return out


class StagePack:
    """Device pointers of one stage's parameters in the C-ABI struct, plus the derived folded tensor `ft_wT`.
    Holds references to every tensor it points to.  Rebuilt when any parameter is replaced or modified in place
    (tracked through `Tensor._version` and `data_ptr`)."""

    def __init__(self, named: dict, device):
        self.keep = []
        self.w = VknStageWeights()
        self.sig = None

        def P(key):
            t = named.get(key)
            if t is None:
                return 0
            t = t.detach()
            if t.device != device or t.dtype != torch.float32 or not t.is_contiguous():
                t = t.to(device=device, dtype=torch.float32).contiguous()
            self.keep.append(t)
            return t.data_ptr()

        w = self.w
        if 'feat_transform.conv.weight' in named:
            ft = named['feat_transform.conv.weight'].detach().to(device=device, dtype=torch.float32)
            ft2 = ft.reshape(ft.shape[0], ft.shape[1]).contiguous()
            ftT = ft2.t().contiguous()
            self.keep += [ft2, ftT]
            w.ft_w, w.ft_wT = ft2.data_ptr(), ftT.data_ptr()
            w.ft_b = P('feat_transform.conv.bias')
        ku = 'kernel_update_conv.'
        w.dyn_w, w.dyn_b = P(ku + 'dynamic_layer.weight'), P(ku + 'dynamic_layer.bias')
        w.inp_w, w.inp_b = P(ku + 'input_layer.weight'), P(ku + 'input_layer.bias')
        w.ig_w, w.ig_b = P(ku + 'input_gate.weight'), P(ku + 'input_gate.bias')
        w.ug_w, w.ug_b = P(ku + 'update_gate.weight'), P(ku + 'update_gate.bias')
        w.norm_in_w, w.norm_in_b = P(ku + 'norm_in.weight'), P(ku + 'norm_in.bias')
        w.norm_out_w, w.norm_out_b = P(ku + 'norm_out.weight'), P(ku + 'norm_out.bias')
        w.inorm_in_w, w.inorm_in_b = P(ku + 'input_norm_in.weight'), P(ku + 'input_norm_in.bias')
        w.inorm_out_w, w.inorm_out_b = P(ku + 'input_norm_out.weight'), P(ku + 'input_norm_out.bias')
        w.fc_w, w.fc_b = P(ku + 'fc_layer.weight'), P(ku + 'fc_layer.bias')
        w.fc_norm_w, w.fc_norm_b = P(ku + 'fc_norm.weight'), P(ku + 'fc_norm.bias')
        w.attn_in_w, w.attn_in_b = P('attention.attn.in_proj_weight'), P('attention.attn.in_proj_bias')
        w.attn_out_w, w.attn_out_b = P('attention.attn.out_proj.weight'), P('attention.attn.out_proj.bias')
        w.attn_norm_w, w.attn_norm_b = P('attention_norm.weight'), P('attention_norm.bias')
        w.ffn1_w, w.ffn1_b = P('ffn.layers.0.0.weight'), P('ffn.layers.0.0.bias')
        w.ffn2_w, w.ffn2_b = P('ffn.layers.1.weight'), P('ffn.layers.1.bias')
        w.ffn_norm_w, w.ffn_norm_b = P('ffn_norm.weight'), P('ffn_norm.bias')
        i = 0
        while f'cls_fcs.{3 * i}.weight' in named:
            w.cls_fc_w[i] = P(f'cls_fcs.{3 * i}.weight')
            w.cls_ln_w[i], w.cls_ln_b[i] = P(f'cls_fcs.{3 * i + 1}.weight'), P(f'cls_fcs.{3 * i + 1}.bias')
            i += 1
        self.n_cls_fcs = i
        i = 0
        while f'mask_fcs.{3 * i}.weight' in named:
            w.mask_fc_w[i] = P(f'mask_fcs.{3 * i}.weight')
            w.mask_ln_w[i], w.mask_ln_b[i] = P(f'mask_fcs.{3 * i + 1}.weight'), P(f'mask_fcs.{3 * i + 1}.bias')
            i += 1
        self.n_mask_fcs = i
        w.fc_cls_w, w.fc_cls_b = P('fc_cls.weight'), P('fc_cls.bias')
        w.fc_mask_w, w.fc_mask_b = P('fc_mask.weight'), P('fc_mask.bias')
        pa = 'attention_previous.attn.'
        w.pa_in_w, w.pa_in_b = P(pa + 'in_proj_weight'), P(pa + 'in_proj_bias')
        w.pa_out_w, w.pa_out_b = P(pa + 'out_proj.weight'), P(pa + 'out_proj.bias')
        w.pa_norm_w, w.pa_norm_b = P('attention_previous_norm.weight'), P('attention_previous_norm.bias')
        w.lffn1_w, w.lffn1_b = P('link_ffn.layers.0.0.weight'), P('link_ffn.layers.0.0.bias')
        w.lffn2_w, w.lffn2_b = P('link_ffn.layers.1.weight'), P('link_ffn.layers.1.bias')
        w.lffn_norm_w, w.lffn_norm_b = P('link_ffn_norm.weight'), P('link_ffn_norm.bias')
        self.has_link = bool(w.pa_in_w and w.lffn1_w)
        self.device = device
        self._prep = None
        self._prep_key = None
        self._ready = None      # event recorded after vkn_prepare_stage_f32: other streams wait on it before using `prepared`

    def ensure_prepared(self, dims):
        """Pre-split every Linear weight into three bf16 terms (vkn_prepare_stage_f32) once per (pack, shape): the
        [N x C] GEMMs then run on bf16 MFMA with fp32-class accuracy instead of exact-fp32 MFMA."""
        key = (dims.C, dims.ff, dims.ncls, dims.n_cls_fcs, dims.n_mask_fcs)
        if self._prep is not None and self._prep_key == key:
            if self._ready is not None and not self._ready.query():
                torch.cuda.current_stream(self.device).wait_event(self._ready)   # prepared on another stream, still in flight
            return
        L = _lib.lib()
        self.w.prepared, self.w.prepared_bytes = None, 0
        nb = L.vkn_prepared_bytes(ctypes.byref(dims), ctypes.byref(self.w))
        if nb == 0:
            return
        buf = torch.empty(nb, dtype=torch.uint8, device=self.device)
        with torch.cuda.device(self.device):
            check(L.vkn_prepare_stage_f32(ctypes.byref(dims), ctypes.byref(self.w), _ptr(buf), nb, _stream()))
            self._ready = torch.cuda.Event()
            self._ready.record(torch.cuda.current_stream(self.device))
        self._prep, self._prep_key = buf, key
        self.w.prepared, self.w.prepared_bytes = buf.data_ptr(), nb

    @staticmethod
    def signature(named: dict, device):
        return (str(device),) + tuple((k, t.data_ptr(), t._version) for k, t in sorted(named.items()))


def link_pack(named: dict, device, updator=None, attention='attention_previous', norm='attention_previous_norm', ffn='link_ffn',
              ffn_norm='link_ffn_norm'):
    """The weights of one previous-frame LINK BLOCK (include/vkn.h: vkn_link_block_f32) as a `StagePack`: `named` are the
    owning stage's parameters, the arguments name the sub-modules the block is made of (`updator` None = no KernelUpdator).
    E.g. previous_link="update_dynamic_cov": updator='attention_previous_update_link', attention='attention_previous_link',
    norm='attention_previous_norm_link', ffn='link_ffn_link', ffn_norm='link_ffn_norm_link'
    (knet/video/kernel_update_head.py:216-236)."""
    remap = {}
    for src, dst in ((updator, 'kernel_update_conv'), (attention, 'attention_previous'), (norm, 'attention_previous_norm'),
                     (ffn, 'link_ffn'), (ffn_norm, 'link_ffn_norm')):
        if src is None:
            continue
        for k, v in named.items():
            if k.startswith(src + '.'):
                remap[dst + k[len(src):]] = v
    return StagePack(remap, device)


def _pw(pack):
    return ctypes.byref(pack.w) if pack is not None else None


def link_block(dims: VknDims, pack: StagePack, cur, prev, update_feature=None):
    """One link block on [B,N,C] kernel sets (see `link_pack`): out = LN(FFN(LN(cur + MHA(cur, kv)))), kv = prev or
    KernelUpdator(update_feature, prev) when the pack carries an updator."""
    cur, prev = _req(cur, 'cur'), _req(prev, 'prev')
    uf = _req(update_feature, 'update_feature') if update_feature is not None else None
    L = _lib.lib()
    pack.ensure_prepared(dims)
    out = torch.empty_like(cur)
    ws = _workspace(max(L.vkn_stage_workspace_bytes(ctypes.byref(dims)), 256), cur.device)
    with torch.cuda.device(cur.device):
        check(L.vkn_link_block_f32(ctypes.byref(dims), ctypes.byref(pack.w), _ptr(uf), _ptr(cur), _ptr(prev), _ptr(out), _ptr(ws),
                                   ws.numel(), _stream()))
    return out


def query_merge(dims: VknDims, pack: StagePack, query, keys, pos=None):
    """Clip-level attention query merge (include/vkn.h: vkn_query_merge_f32): query [B,N,C], keys [B,F*N,C] frame-major, pos [N,C] | None
    -> [B,N,C].  `pack`: `link_pack(named, dev, None, 'query_merge_attn', 'query_merge_norm', 'query_merge_ffn',
    'query_merge_ffn_norm')`; dims.ff is the merge FFN's width."""
    query, keys = _req(query, 'query'), _req(keys, 'keys')
    pos = _req(pos, 'pos') if pos is not None else None
    B, N, C = query.shape
    if keys.shape[0] != B or keys.shape[2] != C or keys.shape[1] % N or (pos is not None and tuple(pos.shape) != (N, C)):
        raise ValueError(f'query {tuple(query.shape)}, keys {tuple(keys.shape)}, pos {None if pos is None else tuple(pos.shape)}')
    F = keys.shape[1] // N
    L = _lib.lib()
    pack.ensure_prepared(dims)
    out = torch.empty_like(query)
    ws = _workspace(max(L.vkn_query_merge_workspace_bytes(ctypes.byref(dims), F), 256), query.device)
    with torch.cuda.device(query.device):
        check(L.vkn_query_merge_f32(ctypes.byref(dims), F, ctypes.byref(pack.w), _ptr(query), _ptr(keys), _ptr(pos), _ptr(out),
                                    _ptr(ws), ws.numel(), _stream()))
    return out


def make_dims(B, N, C, H, W, heads, ff, ncls, n_cls_fcs, n_mask_fcs, hard_mask_thr=0.5, ln_eps=1e-5):
    return VknDims(B, N, C, H, W, heads, ff, ncls, n_cls_fcs, n_mask_fcs, thr_logit(hard_mask_thr), ln_eps)


def stage_forward(dims: VknDims, pack: StagePack, x, obj_in, masks_in, prev_obj=None, want_track=False, flags=0, link_pre=None,
                  link_track=None, track_src=0):
    """One `KernelUpdateHead.forward` on the GPU.  Returns (cls_logits [B,N,ncls], masks [B,N,H,W], obj [B,N,C],
    x_feat [B,N,C], track [B,N,C] | None).  link_pre / link_track: `link_pack`s of the previous_link / previous_type="update"
    blocks (vkn_stage_forward_link_f32)."""
    (x, xdt), obj_in, masks_in = _req_x(x), _req(obj_in, 'proposal_feat'), _req(masks_in, 'mask_preds')
    flags |= (0, FLAG_X_F16, FLAG_X_BF16)[xdt]
    B, N, C, H, W = dims.B, dims.N, dims.C, dims.H, dims.W
    dev = x.device
    L = _lib.lib()
    cls = torch.empty((B, N, dims.ncls), dtype=torch.float32, device=dev)
    masks = torch.empty((B, N, H, W), dtype=torch.float32, device=dev)
    obj = torch.empty((B, N, C), dtype=torch.float32, device=dev)
    xfeat = torch.empty((B, N, C), dtype=torch.float32, device=dev)
    track = None
    if prev_obj is not None and (want_track or link_pre is not None):
        prev_obj = _req(prev_obj, 'previous_obj_feats')
        if want_track:
            track = torch.empty((B, N, C), dtype=torch.float32, device=dev)
    else:
        prev_obj = None
        link_pre = link_track = None
    pack.ensure_prepared(dims)
    for lp in (link_pre, link_track):
        if lp is not None:
            lp.ensure_prepared(dims)
    nb = L.vkn_stage_workspace_bytes(ctypes.byref(dims))  # 0 for unsupported dims: the call below reports the reason
    ws = _workspace(max(nb, 256), dev)
    with torch.cuda.device(dev):
        if link_pre is not None or link_track is not None:
            check(L.vkn_stage_forward_link_f32(ctypes.byref(dims), ctypes.byref(pack.w), _pw(link_pre), _pw(link_track),
                                               int(track_src) if link_track is not None else 0, _ptr(x), _ptr(obj_in), _ptr(masks_in),
                                               _ptr(prev_obj), _ptr(cls), _ptr(masks), _ptr(obj), _ptr(xfeat), _ptr(track),
                                               _ptr(ws), ws.numel(), flags, _stream()))
        else:
            check(L.vkn_stage_forward_f32(ctypes.byref(dims), ctypes.byref(pack.w), _ptr(x), _ptr(obj_in), _ptr(masks_in),
                                          _ptr(prev_obj), _ptr(cls), _ptr(masks), _ptr(obj), _ptr(xfeat), _ptr(track),
                                          _ptr(ws), ws.numel(), flags, _stream()))
    return cls, masks, obj, xfeat, track


def stage_chain(dims: VknDims, pack: StagePack, x_feat, obj_in, want_cls=True, flags=0):
    """The [B*N, C] chain of one stage alone (no gather, no decode): x_feat [B,N,C] (feat-transformed), obj_in [B,N,C] ->
    (cls_logits [B,N,ncls] | None, folded decode kernels [B,N,C], decode bias [B,N], obj [B,N,C])."""
    xf, obj_in = _req(x_feat, 'x_feat'), _req(obj_in, 'proposal_feat')
    B, N, C = dims.B, dims.N, dims.C
    dev = xf.device
    L = _lib.lib()
    cls = torch.empty((B, N, dims.ncls), dtype=torch.float32, device=dev) if want_cls else None
    kern = torch.empty((B, N, C), dtype=torch.float32, device=dev)
    kb = torch.empty((B, N), dtype=torch.float32, device=dev)
    obj = torch.empty((B, N, C), dtype=torch.float32, device=dev)
    pack.ensure_prepared(dims)
    ws = _workspace(max(L.vkn_stage_workspace_bytes(ctypes.byref(dims)), 256), dev)
    with torch.cuda.device(dev):
        check(L.vkn_stage_chain_f32(ctypes.byref(dims), ctypes.byref(pack.w), _ptr(xf), _ptr(obj_in), _ptr(cls), _ptr(kern), _ptr(kb),
                                    _ptr(obj), _ptr(ws), ws.numel(), flags, _stream()))
    return cls, kern, kb, obj


PHASE_A, PHASE_B, PHASE_C = 1024, 2048, 4096     # VKN_FLAG_PHASE_*: include/vkn.h


def head_forward(dims: VknDims, packs, x, proposal_feats, mask_preds, prev_obj=None, upsample_stride=1, want_track=False,
                 want_scaled=True, flags=0, clip_first_prev=None, decode_events=None, link_pre=None, link_track=None, track_src=0,
                 phase=0, out=None):
    """The S-stage loop in one C call.  Returns (obj [B,N,C], cls_prob [B,N,ncls], mask_preds [B,N,H,W],
    scaled_mask_preds [B,N,H*s,W*s] | None, track [B,N,C] | None).  link_pre / link_track / track_src: the LAST stage's
    previous_link / previous_type="update" blocks (`link_pack`; vkn_head_forward_link_f32) — they need prev_obj / clip_first_prev."""
    (x, xdt), pf, mp = _req_x(x), _req(proposal_feats, 'proposal_feats'), _req(mask_preds, 'mask_preds')
    flags |= (0, FLAG_X_F16, FLAG_X_BF16)[xdt]
    B, N, C, H, W = dims.B, dims.N, dims.C, dims.H, dims.W
    dev = x.device
    L = _lib.lib()
    S = len(packs)
    for p in packs:
        p.ensure_prepared(dims)
    arr = (VknStageWeights * S)(*[p.w for p in packs])
    # `phase` (PHASE_A | PHASE_B | PHASE_C, previous_link heads in clip mode only) runs a part of the call; `out` = the tuple a previous
    # phase returned: the later phases write the SAME tensors and continue from the state the earlier ones left in the workspace
    if out is not None:
        obj, cls, masks, scaled, track = out
        scaled = scaled if (want_scaled and upsample_stride > 1) else None
    else:
        obj = torch.empty((B, N, C), dtype=torch.float32, device=dev)
        cls = torch.empty((B, N, dims.ncls), dtype=torch.float32, device=dev)
        masks = torch.empty((B, N, H, W), dtype=torch.float32, device=dev)
        scaled = None
        if want_scaled and upsample_stride > 1:
            scaled = torch.empty((B, N, H * upsample_stride, W * upsample_stride),
                                 dtype=torch.float16 if (flags & FLAG_SCALED_F16) else torch.float32, device=dev)
        track = None
    flags |= int(phase)
    if clip_first_prev is not None:
        # the B frames are consecutive frames of one video: frame b links to frame b - 1 of this call, frame 0 to `clip_first_prev`
        # [1,N,C] (VKN_FLAG_CLIP_LINK) — the whole clip step is one C call
        prev_obj = _req(clip_first_prev.reshape(1, N, C), 'clip_first_prev')
        if track is None:
            track = torch.empty((B, N, C), dtype=torch.float32, device=dev)
        flags |= 8
    elif prev_obj is not None and (want_track or link_pre is not None):
        prev_obj = _req(prev_obj, 'previous_obj_feats')
        if want_track:
            track = torch.empty((B, N, C), dtype=torch.float32, device=dev)
    else:
        prev_obj = None
        link_pre = link_track = None
    for lp in (link_pre, link_track):
        if lp is not None:
            lp.ensure_prepared(dims)
    nb = L.vkn_head_workspace_bytes(ctypes.byref(dims))
    ws = _workspace(max(nb, 256), dev)
    with torch.cuda.device(dev):
        if link_pre is not None or link_track is not None:
            check(L.vkn_head_forward_link_f32(ctypes.byref(dims), S, arr, _pw(link_pre), _pw(link_track),
                                              int(track_src) if link_track is not None else 0, _ptr(x), _ptr(pf), _ptr(mp),
                                              _ptr(prev_obj), _ptr(obj), _ptr(cls), _ptr(masks), _ptr(scaled), int(upsample_stride),
                                              _ptr(track), _ptr(ws), ws.numel(), flags, _stream()))
        elif decode_events is not None:
            # (start, stop) torch.cuda.Event(enable_timing=True) pair, each recorded once before (that creates the HIP event): the
            # library records them around the last stage's mask-decode launch on the current stream
            e0, e1 = decode_events
            check(L.vkn_head_forward_prof_f32(ctypes.byref(dims), S, arr, _ptr(x), _ptr(pf), _ptr(mp), _ptr(prev_obj), _ptr(obj),
                                              _ptr(cls), _ptr(masks), _ptr(scaled), int(upsample_stride), _ptr(track), _ptr(ws),
                                              ws.numel(), flags, _stream(), ctypes.c_void_p(e0.cuda_event),
                                              ctypes.c_void_p(e1.cuda_event)))
        else:
            check(L.vkn_head_forward_f32(ctypes.byref(dims), S, arr, _ptr(x), _ptr(pf), _ptr(mp), _ptr(prev_obj), _ptr(obj),
                                         _ptr(cls), _ptr(masks), _ptr(scaled), int(upsample_stride), _ptr(track), _ptr(ws),
                                         ws.numel(), flags, _stream()))
    return obj, cls, masks, (scaled if scaled is not None else masks), track


def split_planes(kernels):
    """fp32 kernels [B,N,C] -> (hi, lo) f16 planes [B, roundup(N,32), C] with hi + lo ~= kernels (2^-22 relative)."""
    k = _req(kernels.reshape(kernels.shape[0], kernels.shape[1], -1), 'kernels')
    B, N, C = k.shape
    npt = (N + 31) // 32 * 32
    hi = torch.zeros((B, npt, C), dtype=torch.float16, device=k.device)
    lo = torch.zeros((B, npt, C), dtype=torch.float16, device=k.device)
    with torch.cuda.device(k.device):
        check(_lib.lib().vkn_split_planes_f32(_ptr(k), _ptr(hi), _ptr(lo), B, N, C, _stream()))
    return hi, lo


def mask_decode_planes(x, hi, lo, N, bias=None, out=None):
    """The MFMA decode kernel alone on pre-split kernel planes (what runs inside a stage); `out` may be preallocated."""
    x, xdt = _req_x(x)
    B, C, H, W = x.shape
    if out is None:
        out = torch.empty((B, N, H, W), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        check(_lib.lib().vkn_mask_decode_planes_x(_ptr(x), xdt, _ptr(hi), _ptr(lo), _ptr(bias), _ptr(out), B, N, C, H * W,
                                                  _stream()))
    return out


def decode_gather(x, hi, lo, N, bias=None, hard_mask_thr=0.5):
    """Stage s decode fused with the stage s + 1 gather, one pass over x: (xraw [B,N,C], cnt [B,N]) of the masks
    `bias + K.x >= thr` without materialising them (bit-identical to mask_decode_planes + mask_gather)."""
    x, xdt = _req_x(x)
    B, C, H, W = x.shape
    P = H * W
    L = _lib.lib()
    xraw = torch.empty((B, N, C), dtype=torch.float32, device=x.device)
    cnt = torch.empty((B, N), dtype=torch.float32, device=x.device)
    ws = _workspace(L.vkn_gather_workspace_bytes(B, N, C, P), x.device, scratch=True)
    with torch.cuda.device(x.device):
        check(L.vkn_decode_gather_x(_ptr(x), xdt, _ptr(hi), _ptr(lo), _ptr(bias), thr_logit(hard_mask_thr), _ptr(xraw), _ptr(cnt),
                                    B, N, C, P, _ptr(ws), ws.numel(), _stream()))
    return xraw, cnt


def track_link(dims: VknDims, pack: StagePack, cur_obj, prev_obj, flags=0):
    """Tracking embedding of the video head's last stage for a batch of (cur, prev) kernel sets:
    knet/video/kernel_update_head.py:394-415.  cur_obj, prev_obj [B,N,C] -> [B,N,C].  `flags`: FLAG_CHAIN_KSPLIT / _LAUNCHES pin the
    link's arithmetic to the form a larger call took (a hand-over re-link of one frame; include/vkn.h: vkn_track_link_flags_f32)."""
    cur, prev = _req(cur_obj, 'cur_obj'), _req(prev_obj, 'prev_obj')
    L = _lib.lib()
    pack.ensure_prepared(dims)
    out = torch.empty_like(cur)
    ws = _workspace(max(L.vkn_stage_workspace_bytes(ctypes.byref(dims)), 256), cur.device)
    with torch.cuda.device(cur.device):
        check(L.vkn_track_link_flags_f32(ctypes.byref(dims), ctypes.byref(pack.w), _ptr(cur), _ptr(prev), _ptr(out), _ptr(ws),
                                         ws.numel(), int(flags), _stream()))
    return out


def split_weight(W):
    """fp32 Linear weight [Nout, K] -> bf16x3 tile images (opaque uint8 buffer, 6 * roundup(Nout,256) * K bytes) for
    `linear(..., w_split=...)`."""
    W = _req(W, 'W')
    Nout, K = W.shape
    buf = torch.empty(6 * ((Nout + 255) // 256 * 256) * K, dtype=torch.uint8, device=W.device)   # 256-row tile images
    with torch.cuda.device(W.device):
        check(_lib.lib().vkn_split_weight_f32(_ptr(W), _ptr(buf), Nout, K, _stream()))
    return buf


def linear(A, W, bias=None, w_split=None, act=0, ksplit=1):
    """act(A @ W.T + bias) through the library's GEMM kernel (exact-fp32 MFMA, or bf16x3 split MFMA when `w_split` is given)."""
    A, W = _req(A, 'A'), _req(W, 'W')
    M, K = A.shape
    Nout = W.shape[0]
    out = torch.empty((M, Nout), dtype=torch.float32, device=A.device)
    ws = _workspace(max(ksplit * M * Nout * 4, 256), A.device, scratch=True)
    with torch.cuda.device(A.device):
        check(_lib.lib().vkn_linear_f32(_ptr(A), _ptr(W), _ptr(w_split), _ptr(bias), _ptr(out), M, K, Nout, int(act), int(ksplit),
                                        _ptr(ws), ws.numel(), _stream()))
    return out


def kernel_init(loc_feats, semantic_feats, init_w, seg_w=None, seg_b=None, num_thing_classes=0, cat_stuff_mask=False,
                proposal_feats_with_obj=True, hard_mask_thr=0.5, want_seg_preds=True, flags=0, use_binary=True):
    """Kernel initialisation ("pass 0"), `ConvKernelHead._decode_init_proposals` after its loc / seg convs
    (knet/det/kernel_head.py:204-263), use_binary semantics.  Returns (proposal_feats [B,N,C], x_feats [B,C,H,W],
    mask_preds [B,N,H,W], seg_preds [B,ncls,H,W] | None)."""
    loc, xdt = _req_x(loc_feats, 'loc_feats')       # fp32, or fp16 / bf16 STORAGE (x_feats then has the same type; the head reads it as is)
    flags = int(flags) | {1: FLAG_X_F16, 2: FLAG_X_BF16}.get(xdt, 0)
    B, C, H, W = loc.shape
    P = H * W
    dev = loc.device
    iw = _req(init_w.reshape(init_w.shape[0], -1), 'init_kernels.weight')
    Np = iw.shape[0]
    if iw.shape[1] != C:
        raise ValueError('init_kernels.weight must be [num_proposals, C, 1, 1] (conv_kernel_size == 1)')
    sem = sw = sb = None
    ncls = 0
    if semantic_feats is not None:
        sem, sdt = _req_x(semantic_feats, 'semantic_feats')
        if sem.shape != loc.shape or sdt != xdt:
            raise ValueError('semantic_feats and loc_feats must have the same shape and storage type')
        sw = _req(seg_w.reshape(seg_w.shape[0], -1), 'conv_seg.weight')
        sb = _req(seg_b, 'conv_seg.bias') if seg_b is not None else None
        ncls = sw.shape[0]
    elif cat_stuff_mask:
        raise ValueError('cat_stuff_mask needs the semantic branch')
    nstuff = ncls - num_thing_classes if cat_stuff_mask else 0
    N = Np + nstuff
    L = _lib.lib()
    x_feats = torch.empty_like(loc) if sem is not None else loc
    masks = torch.empty((B, N, H, W), dtype=torch.float32, device=dev)
    seg = torch.empty((B, ncls, H, W), dtype=torch.float32, device=dev) if (sem is not None and want_seg_preds) else None
    prop = torch.empty((B, N, C), dtype=torch.float32, device=dev)
    nb = L.vkn_kernel_init_workspace_bytes(B, Np, ncls, C, P)
    ws = _workspace(max(nb, 256), dev, scratch=True)
    with torch.cuda.device(dev):
        check(L.vkn_kernel_init_f32(_ptr(loc), _ptr(sem), _ptr(iw), _ptr(sw), _ptr(sb), int(num_thing_classes),
                                    int(bool(cat_stuff_mask)), (1 if use_binary else 2) if proposal_feats_with_obj else 0,
                                    thr_logit(hard_mask_thr),
                                    _ptr(x_feats), _ptr(masks), _ptr(seg), _ptr(prop), B, Np, ncls, C, P, _ptr(ws), ws.numel(),
                                    flags, _stream()))
    return prop, x_feats, masks, seg


def panoptic_joint(cls_prob, mask_logits, num_proposals, num_thing_classes, max_per_img, instance_score_thr, overlap_thr,
                   img_shape, batch_input_shape, ori_shape, upsample_stride=1, want_bbox=False):
    """Joint panoptic merge of a batch of frames sharing one img_meta, straight from the head's low-res mask logits
    (`get_panoptic` + `merge_stuff_thing_stuff_joint` + `rescale_masks`, knet/det/kernel_iter_head.py:332-370, 467-524).
    Returns device tensors (panoptic_seg int32 [B,Ho,Wo], info int32 [B,K,6], nseg int32 [B]) and, with `want_bbox`, a fourth
    one: bbox int32 [B,K,4] = (xmin, ymin, xmax, ymax) of every accepted segment (what the tracker consumes); see include/vkn.h."""
    cls, m = _req(cls_prob, 'cls_prob'), _req(mask_logits, 'mask_logits')
    B, N, ncls = cls.shape
    if m.shape[0] != B or m.shape[1] != N:
        raise ValueError('cls_prob [B,N,ncls] and mask_logits [B,N,Hm,Wm] disagree')
    Hm, Wm = int(m.shape[2]), int(m.shape[3])
    cfg = _lib.VknPanopticCfg(int(num_proposals), int(num_thing_classes), int(max_per_img), float(instance_score_thr),
                              float(overlap_thr), int(upsample_stride), Hm, Wm, int(batch_input_shape[0]),
                              int(batch_input_shape[1]), int(img_shape[0]), int(img_shape[1]), int(ori_shape[0]),
                              int(ori_shape[1]))
    K = int(max_per_img) + (N - int(num_proposals))
    dev = cls.device
    L = _lib.lib()
    seg = torch.empty((B, cfg.Ho, cfg.Wo), dtype=torch.int32, device=dev)
    info = torch.empty((B, K, 6), dtype=torch.int32, device=dev)
    nseg = torch.empty((B,), dtype=torch.int32, device=dev)
    bbox = torch.empty((B, K, 4), dtype=torch.int32, device=dev) if want_bbox else None
    nb = L.vkn_panoptic_workspace_bytes(ctypes.byref(cfg), B, N)
    ws = _workspace(max(nb, 256), dev, scratch=True)
    with torch.cuda.device(dev):
        check(L.vkn_panoptic_joint_f32(ctypes.byref(cfg), _ptr(cls), _ptr(m), B, N, ncls, seg.data_ptr(), info.data_ptr(),
                                       nseg.data_ptr(), bbox.data_ptr() if want_bbox else None, _ptr(ws), ws.numel(), _stream()))
    return (seg, info, nseg, bbox) if want_bbox else (seg, info, nseg)


def panoptic_thing_first(thing_masks, thing_scores, thing_labels, thing_order, stuff_masks, stuff_labels, stuff_order,
                         instance_score_thr, iou_thr, stuff_max_area):
    """Thing-first panoptic merge of ONE image on the device (`merge_stuff_thing`, knet/det/kernel_iter_head.py:385-465).
    thing_masks [Kt,H,W] / stuff_masks [Ks,H,W] bool, scores fp32, labels / orders int; `*_order` = the paste order (indices into
    the mask arrays).  Returns device tensors (panoptic_seg int32 [H,W], info int32 [Kt+Ks,5], nseg int32 [1]); see include/vkn.h."""
    dev = thing_masks.device
    if not thing_masks.is_cuda:
        raise _lib.VknLibraryError('panoptic_thing_first: expected CUDA/HIP tensors — the MI355X path has no CPU fallback')
    H, W = thing_masks.shape[-2:]
    HW = H * W
    Kt, Ks = thing_masks.shape[0], stuff_masks.shape[0]
    tm = thing_masks.to(torch.uint8).contiguous()
    sm = stuff_masks.to(torch.uint8).contiguous()
    i32 = lambda t: t.to(device=dev, dtype=torch.int32).contiguous()  # noqa: E731
    ts, tl, to, sl, so = thing_scores.float().contiguous(), i32(thing_labels), i32(thing_order), i32(stuff_labels), i32(stuff_order)
    L = _lib.lib()
    seg = torch.empty((H, W), dtype=torch.int32, device=dev)
    info = torch.zeros((Kt + Ks, 5), dtype=torch.int32, device=dev)
    nseg = torch.zeros((1,), dtype=torch.int32, device=dev)
    ws = _workspace(max(L.vkn_merge_workspace_bytes(Kt, Ks), 256), dev, scratch=True)
    with torch.cuda.device(dev):
        check(L.vkn_panoptic_thing_first_u8(_ptr(tm), _ptr(ts), tl.data_ptr(), to.data_ptr(), Kt, _ptr(sm), sl.data_ptr(), so.data_ptr(),
                                            Ks, HW, float(instance_score_thr), float(iou_thr), int(stuff_max_area), seg.data_ptr(),
                                            info.data_ptr(), nseg.data_ptr(), _ptr(ws), ws.numel(), _stream()))
    return seg, info, nseg


def assign_costs(mask_logits, cls_logits, gt_masks, gt_labels, cls_weight=2.0, dice_weight=4.0, mask_weight=1.0,
                 focal_alpha=0.25, focal_gamma=2.0, focal_eps=1e-12, dice_eps=1e-3, dice_pred_min=1e-3, mask_pred_min=1e-2,
                 labels_checked=False):
    """Cost matrix [N, G] of `MaskHungarianAssigner.assign` (knet/det/mask_hungarian_assigner.py:222-241) on the GPU.
    dice_pred_min / mask_pred_min: the lower clamp of sigmoid(logits) in DiceCost / MaskCost (knet: 1e-3 / 1e-2; knet_vis: none)."""
    m = _req(mask_logits.reshape(mask_logits.shape[0], -1), 'mask_preds')
    g = _req(gt_masks.reshape(gt_masks.shape[0], -1).float(), 'gt_masks')
    N, P = m.shape
    G = g.shape[0]
    if g.shape[1] != P:
        raise ValueError('mask_preds and gt_masks must have the same spatial size')
    cls = _req(cls_logits, 'cls_pred') if cls_logits is not None else None
    ncls = cls.shape[1] if cls is not None else 0
    lab = gt_labels.to(device=m.device, dtype=torch.int32).contiguous()
    if cls is not None and lab.numel() and not labels_checked:
        # the reference's `cls_pred[:, gt_labels]` raises an IndexError for labels outside the logits (ignore label 255, stuff label
        # against thing-only logits); an unchecked device read would silently produce garbage costs.  Host labels are checked on the
        # host; device labels cost ONE combined read (the cost matrix itself goes to the host for the LSAP right after)
        src = gt_labels if not gt_labels.is_cuda else torch.stack(torch.aminmax(lab))
        lo, hi = (int(v) for v in (src.min(), src.max())) if not gt_labels.is_cuda else (int(v) for v in src.tolist())
        if lo < 0 or hi >= ncls:
            raise IndexError(f'gt_labels outside [0, {ncls}): {lo} .. {hi}')
    cfg = _lib.VknAssignCfg(float(cls_weight), float(dice_weight), float(mask_weight), float(focal_alpha), float(focal_gamma),
                            float(focal_eps), float(dice_eps), float(dice_pred_min), float(mask_pred_min))
    L = _lib.lib()
    cost = torch.empty((N, G), dtype=torch.float32, device=m.device)
    nb = L.vkn_assign_workspace_bytes(N, G, P)
    ws = _workspace(max(nb, 256), m.device, scratch=True)
    with torch.cuda.device(m.device):
        check(L.vkn_assign_costs_f32(ctypes.byref(cfg), _ptr(m), _ptr(cls), _ptr(g), lab.data_ptr(), N, G, ncls, P, _ptr(cost),
                                     _ptr(ws), ws.numel(), _stream()))
    return cost


_LABS32 = {}


def _labels_i32(gt_labels, dev):
    """the batch's labels as ONE int32 tensor — the same tensors arrive once per stage: built once per (tensor identities, versions)"""
    import weakref
    key = tuple((l.data_ptr(), l._version, tuple(l.shape), str(l.device)) for l in gt_labels) + (str(dev),)
    hit = _LABS32.get(key)
    if hit is not None and all(r() is l for r, l in zip(hit[0], gt_labels)):
        return hit[1]
    if len(_LABS32) > 16:
        _LABS32.clear()
    labs = torch.cat([l.reshape(-1) for l in gt_labels]).to(device=dev, dtype=torch.int32)
    _LABS32[key] = ([weakref.ref(l) for l in gt_labels], labs)
    return labs


def assign_costs_batch(mask_logits, cls_logits, gt_masks, gt_labels, cls_weight=2.0, dice_weight=4.0, mask_weight=1.0,
                       focal_alpha=0.25, focal_gamma=2.0, focal_eps=1e-12, dice_eps=1e-3, dice_pred_min=1e-3, mask_pred_min=1e-2):
    """`assign_costs` for the images of a batch in ONE C call (vkn_assign_costs_batch_f32): lists of per-image [N, H, W] logits,
    [N, ncls] class logits (or None for all), [G_b, H, W] float ground-truth masks and [G_b] labels — ALREADY range-checked (the
    per-image entry point checks; this one is the training loop's).  -> list of [N, G_b] cost matrices (views of one allocation)."""
    n = len(mask_logits)
    N = mask_logits[0].shape[0]
    P = mask_logits[0][0].numel()
    dev = mask_logits[0].device
    use_cls = cls_logits is not None and cls_logits[0] is not None and cls_weight != 0
    ncls = cls_logits[0].shape[1] if use_cls else 0
    Gs = [int(g.shape[0]) for g in gt_masks]
    labs = _labels_i32(gt_labels, dev) if use_cls else None
    cost = torch.empty((N * sum(Gs),), dtype=torch.float32, device=dev)
    probs = (_lib.VknAssignProblem * n)()
    keep, out, off = [], [], 0
    for b in range(n):
        m = _req(mask_logits[b].reshape(N, P), 'mask_preds')
        g = gt_masks[b].reshape(Gs[b], -1)
        if g.dtype != torch.float32:
            g = g.float()
        g = _req(g, 'gt_masks')
        if g.shape[1] != P or m.shape[0] != N:
            raise ValueError('mask_preds and gt_masks must have the same spatial size, and every image the same number of predictions')
        c = _req(cls_logits[b], 'cls_pred') if use_cls else None
        cb = cost[N * off:N * (off + Gs[b])].view(N, Gs[b])
        probs[b] = _lib.VknAssignProblem(m.data_ptr(), c.data_ptr() if use_cls else None, g.data_ptr(),
                                         labs.data_ptr() + 4 * off if use_cls else None, Gs[b], cb.data_ptr())
        keep += [m, g, c]
        out.append(cb)
        off += Gs[b]
    cfg = _lib.VknAssignCfg(float(cls_weight if use_cls else 0.0), float(dice_weight), float(mask_weight), float(focal_alpha),
                            float(focal_gamma), float(focal_eps), float(dice_eps), float(dice_pred_min), float(mask_pred_min))
    L = _lib.lib()
    ws = _workspace(max(L.vkn_assign_workspace_bytes(N, max(Gs), P), 256), dev, scratch=True)
    with torch.cuda.device(dev):
        check(L.vkn_assign_costs_batch_f32(ctypes.byref(cfg), probs, n, N, ncls, P, _ptr(ws), ws.numel(), _stream()))
    return out


def assign_costs_lowres_supported(N, Gs, h, w, stride):
    """Does `assign_costs_lowres_batch` take this shape?  (include/vkn.h: vkn_assign_costs_lowres_batch_f32)"""
    return bool(Gs) and len(Gs) <= 16 and min(Gs) > 0 and \
        _lib.lib().vkn_assign_lowres_workspace_bytes(len(Gs), int(N), max(Gs), int(h), int(w), int(stride)) > 0


def assign_costs_lowres_batch(low_logits, stride, cls_logits, gt_masks, gt_labels, cls_weight=2.0, dice_weight=4.0, mask_weight=1.0,
                              focal_alpha=0.25, focal_gamma=2.0, focal_eps=1e-12, dice_eps=1e-3, dice_pred_min=1e-3, mask_pred_min=1e-2):
    """`assign_costs_batch` on the LOW-RES logits: per-image [N, h, w] logits whose x`stride` bilinear up-scaling the reference assigns
    on, [G_b, stride h, stride w] ground truths — interpolation, activation and contraction in one kernel for the whole batch
    (vkn_assign_costs_lowres_batch_f32: the up-scaled predictions are never read).  Labels ALREADY range-checked.
    -> list of [N, G_b] cost matrices (views of one allocation)."""
    n = len(low_logits)
    N, h, w = (int(v) for v in low_logits[0].shape)
    dev = low_logits[0].device
    use_cls = cls_logits is not None and cls_logits[0] is not None and cls_weight != 0
    ncls = cls_logits[0].shape[1] if use_cls else 0
    Gs = [int(g.shape[0]) for g in gt_masks]
    labs = _labels_i32(gt_labels, dev) if use_cls else None
    cost = torch.empty((N * sum(Gs),), dtype=torch.float32, device=dev)
    probs = (_lib.VknAssignProblem * n)()
    keep, out, off = [], [], 0
    for b in range(n):
        m = _req(low_logits[b], 'mask_preds')
        g = gt_masks[b]
        if g.dtype != torch.float32:
            g = g.float()
        g = _req(g, 'gt_masks')
        if tuple(m.shape) != (N, h, w) or tuple(g.shape[1:]) != (stride * h, stride * w):
            raise ValueError('every image needs [N, h, w] logits and [G, stride h, stride w] ground-truth masks')
        c = _req(cls_logits[b], 'cls_pred') if use_cls else None
        cb = cost[N * off:N * (off + Gs[b])].view(N, Gs[b])
        probs[b] = _lib.VknAssignProblem(m.data_ptr(), c.data_ptr() if use_cls else None, g.data_ptr(),
                                         labs.data_ptr() + 4 * off if use_cls else None, Gs[b], cb.data_ptr())
        keep += [m, g, c]
        out.append(cb)
        off += Gs[b]
    cfg = _lib.VknAssignCfg(float(cls_weight if use_cls else 0.0), float(dice_weight), float(mask_weight), float(focal_alpha),
                            float(focal_gamma), float(focal_eps), float(dice_eps), float(dice_pred_min), float(mask_pred_min))
    L = _lib.lib()
    nb = L.vkn_assign_lowres_workspace_bytes(n, N, max(Gs), h, w, int(stride))
    if nb == 0:
        raise ValueError('shape outside vkn_assign_costs_lowres_batch_f32 (ops.assign_costs_lowres_supported)')
    ws = _workspace(max(nb, 256), dev, scratch=True)
    with torch.cuda.device(dev):
        check(L.vkn_assign_costs_lowres_batch_f32(ctypes.byref(cfg), probs, n, N, ncls, h, w, int(stride), _ptr(ws), ws.numel(), _stream()))
    return out


def focal_loss_fwd(logits, labels, row_weight, alpha, gamma):
    """Sigmoid focal loss in one pass (include/vkn.h: vkn_focal_loss_f32).  logits [M, ncls] fp32, labels int64 [M], row_weight [M] |
    [M, ncls] | None -> (sum of the weighted element losses: 0-d tensor, d sum / d logits [M, ncls])."""
    z = _req(logits, 'cls_score')
    M, ncls = z.shape
    lab = labels.to(device=z.device, dtype=torch.int64).contiguous()
    ew = row_weight is not None and row_weight.numel() == M * ncls and ncls > 1
    w = None
    if row_weight is not None:
        w = _req(row_weight.reshape((M, ncls) if ew else (M,)).float(), 'label_weights')
    L = _lib.lib()
    part = torch.empty((L.vkn_focal_loss_blocks(M, ncls),), dtype=torch.float32, device=z.device)
    grad = torch.empty_like(z)
    with torch.cuda.device(z.device):
        check(L.vkn_focal_loss_f32(_ptr(z), lab.data_ptr(), _ptr(w), int(ew), M, ncls, float(alpha), float(gamma), _ptr(part), _ptr(grad),
                                   _stream()))
    return part.sum(), grad


def mask_losses_fwd(pred, target, pos_rows, rowk, B, with_rank):
    """Forward sums of the three mask losses (include/vkn.h: vkn_mask_losses_fwd_f32).  pred, target [R, P]; pos_rows int64 [K]; rowk
    int32 [R].  -> (rowstats [K, 4] = (sum bce, sum p t, sum p^2, sum t^2), lse [B, P] | None, top int32 [B, P] | None,
    rank_sum scalar tensor | None)."""
    pred, target = _req(pred, 'pred'), _req(target, 'target')
    R, P = pred.shape
    K, Ns = int(pos_rows.shape[0]), R // B
    L = _lib.lib()
    nch, nbl = L.vkn_mask_losses_chunks(P), L.vkn_mask_losses_blocks(P)
    dev = pred.device
    rp = torch.empty((K, nch, 4), dtype=torch.float32, device=dev)
    lse = torch.empty((B, P), dtype=torch.float32, device=dev) if with_rank else None
    top = torch.empty((B, P), dtype=torch.int32, device=dev) if with_rank else None
    rkp = torch.empty((B, nbl), dtype=torch.float32, device=dev) if with_rank else None
    with torch.cuda.device(dev):
        check(L.vkn_mask_losses_fwd_f32(_ptr(pred), _ptr(target), pos_rows.data_ptr() if K else None, rowk.data_ptr(), K, B, Ns, P,
                                        1 if with_rank else 0, _ptr(rp) if K else None, _ptr(lse), top.data_ptr() if with_rank else None,
                                        _ptr(rkp), _stream()))
    return rp.sum(dim=1), lse, top, (rkp.sum() if with_rank else None)


def mask_losses_bwd(pred, target, rowk, rowcoef, coef, lse, top, B, with_rank):
    """Gradient of the three mask losses w.r.t. pred [R, P] in one pass (vkn_mask_losses_bwd_f32)."""
    pred, target = _req(pred, 'pred'), _req(target, 'target')
    R, P = pred.shape
    grad = torch.empty_like(pred)
    with torch.cuda.device(pred.device):
        check(_lib.lib().vkn_mask_losses_bwd_f32(_ptr(pred), _ptr(target), rowk.data_ptr(), _ptr(rowcoef), _ptr(coef), _ptr(lse),
                                                 top.data_ptr() if with_rank else None, B, R // B, P, 1 if with_rank else 0,
                                                 _ptr(grad), _stream()))
    return grad


_POW2_SCRATCH = {}


def pow2_scale(t, target_log2=10):
    """-> device float32 [8] with [0] = the power of two s putting max|t| into [2^(target_log2 - 1), 2^target_log2) and [4] = 1 / s
    (vkn_pow2_scale_f32: one launch, nothing read on the host).  `scale_of(s8)` / `inv_of(s8)` give the two as 16-byte aligned views."""
    t = _req(t.detach(), 't')
    dev = t.device
    key = (dev, torch.cuda.current_stream(dev).cuda_stream)
    scr = _POW2_SCRATCH.get(key)
    if scr is None:
        scr = _POW2_SCRATCH[key] = torch.zeros(2, dtype=torch.int32, device=dev)     # the kernel leaves it zero again
    out = torch.empty(8, dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        check(_lib.lib().vkn_pow2_scale_f32(_ptr(t), t.numel(), int(target_log2), _ptr(out), scr.data_ptr(), _stream()))
    return out


def scale_of(s8):
    return s8[0:1]


def inv_of(s8):
    return s8[4:5]


def scale_pad_rows(t, scale, mult=32):
    """t [B, R, ...] * scale (device scalar or None) with the rows zero-padded to a multiple of `mult` (vkn_scale_pad_rows_f32)."""
    t = _req(t, 't')
    B, R = t.shape[:2]
    P = t[0, 0].numel()
    Rp = (R + mult - 1) // mult * mult
    out = torch.empty((B, Rp) + tuple(t.shape[2:]), dtype=torch.float32, device=t.device)
    with torch.cuda.device(t.device):
        check(_lib.lib().vkn_scale_pad_rows_f32(_ptr(t), _ptr(scale), B, R, Rp, P, _ptr(out), _stream()))
    return out


def transpose_pad(k, scale, Np):
    """k [B, N, C] -> [B, C, Np] = k^T * scale (device scalar or None), columns N .. Np zero (vkn_transpose_pad_f32)."""
    k = _req(k, 'k')
    B, N, C = k.shape
    out = torch.empty((B, C, Np), dtype=torch.float32, device=k.device)
    with torch.cuda.device(k.device):
        check(_lib.lib().vkn_transpose_pad_f32(_ptr(k), _ptr(scale), B, N, C, int(Np), _ptr(out), _stream()))
    return out


def threshold_rows_f16(mask_logits, hard_mask_thr, mult=32):
    """[B, N, H, W] logits -> fp16 [B, Np, H, W] = (logit >= thr_logit(hard_mask_thr)), rows padded with zeros to a multiple of `mult`."""
    m = _req(mask_logits, 'mask_logits')
    B, N = m.shape[:2]
    P = m[0, 0].numel()
    Np = (N + mult - 1) // mult * mult
    out = torch.empty((B, Np) + tuple(m.shape[2:]), dtype=torch.float16, device=m.device)
    with torch.cuda.device(m.device):
        check(_lib.lib().vkn_threshold_rows_f16(_ptr(m), thr_logit(hard_mask_thr), B, N, Np, P, _ptr(out), _stream()))
    return out


def unscale_rows(dk_p, dkb_p, scale, N):
    """(dk_p [B, Np, C], dkb_p [B, Np]) -> (dk [B, N, C], dkb [B, N]) = the first N rows times the device scalar (vkn_unscale_rows_f32)."""
    dk_p, dkb_p = _req(dk_p, 'dk'), _req(dkb_p, 'dkb')
    B, Np, C = dk_p.shape
    dk = torch.empty((B, N, C), dtype=torch.float32, device=dk_p.device)
    dkb = torch.empty((B, N), dtype=torch.float32, device=dk_p.device)
    with torch.cuda.device(dk_p.device):
        check(_lib.lib().vkn_unscale_rows_f32(_ptr(dk_p), _ptr(dkb_p), _ptr(scale), B, int(N), Np, C, _ptr(dk), _ptr(dkb), _stream()))
    return dk, dkb


SUM_MAX = 8


def sum_tensors(parts):
    """parts[0] + parts[1] + ... (same shape; in that order) in ONE pass for fp32 CUDA tensors (vkn_sum_n_f32), else torch adds."""
    if len(parts) == 1:
        return parts[0]
    p0 = parts[0]
    if not (p0.is_cuda and all(p.dtype == torch.float32 and p.shape == p0.shape and p.device == p0.device for p in parts)):
        out = parts[0] + parts[1]
        for p in parts[2:]:
            out = out + p
        return out
    out = None
    while len(parts) > 1:
        take = [_req(p, 'part') for p in parts[:SUM_MAX]]
        arr = (ctypes.c_void_p * len(take))(*[p.data_ptr() for p in take])
        out = torch.empty_like(take[0])
        with torch.cuda.device(p0.device):
            check(_lib.lib().vkn_sum_n_f32(arr, len(take), out.numel(), _ptr(out), _stream()))
        parts = [out] + list(parts[SUM_MAX:])
    return out


def check_range(values, lo, hi, flag, status):
    """*status (device int32 [1]) |= flag when an element of the integer tensor `values` lies outside [lo, hi) (vkn_check_range_i64)."""
    v = values.reshape(-1)
    if v.dtype != torch.int64 or not v.is_contiguous():
        v = v.to(torch.int64).contiguous()
    with torch.cuda.device(v.device):
        check(_lib.lib().vkn_check_range_i64(v.data_ptr(), v.numel(), int(lo), int(hi), int(flag), status.data_ptr(), _stream()))


def lsap_device(costs):
    """`scipy.optimize.linear_sum_assignment` for a batch of DEVICE cost matrices [nr_b, nc_b] (fp32) without leaving the device:
    one launch, one wavefront per matrix (include/vkn.h: vkn_lsap_batch_f32).  -> (gt_inds, row_ind, col_ind, status):
    per matrix gt_inds int64 [nr] (matched column + 1, 0 = unmatched: the reference's `assigned_gt_inds`), the min(nr, nc) matched
    (row, col) pairs sorted by row (int32), and ONE int32 status tensor [batch] (0 ok, 1 invalid entries, 2 infeasible) that
    the caller may read whenever it wants to pay for the synchronisation."""
    if not costs:
        return [], [], [], None
    dev = costs[0].device
    probs = (_lib.VknLsapProblem * len(costs))()
    gts, rows, cols, keep = [], [], [], []
    for b, c in enumerate(costs):
        c = _req(c, 'cost')
        if c.dim() != 2 or c.shape[0] == 0 or c.shape[1] == 0:
            raise ValueError(f'cost {b}: expected a non-empty 2-D matrix, got {tuple(c.shape)}')
        nr, nc = c.shape
        k = min(nr, nc)
        g = torch.empty(nr, dtype=torch.int64, device=dev)
        r = torch.empty(k, dtype=torch.int32, device=dev)
        cc = torch.empty(k, dtype=torch.int32, device=dev)
        probs[b] = _lib.VknLsapProblem(c.data_ptr(), nr, nc, g.data_ptr(), r.data_ptr(), cc.data_ptr())
        gts.append(g); rows.append(r); cols.append(cc); keep.append(c)
    status = torch.empty(len(costs), dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        check(_lib.lib().vkn_lsap_batch_f32(probs, len(costs), status.data_ptr(), _stream()))
    for c in keep:                       # the launch is asynchronous: the matrices must outlive it on this stream
        c.record_stream(torch.cuda.current_stream(dev))
    return gts, rows, cols, status


def lsap(cost):
    """`scipy.optimize.linear_sum_assignment(cost)` on a host fp32 matrix through libvkn's C++ solver -> (row_ind, col_ind)
    int64 numpy arrays (knet/det/mask_hungarian_assigner.py:244-251)."""
    import numpy as np
    c = np.ascontiguousarray(cost.detach().cpu().numpy() if torch.is_tensor(cost) else cost, dtype=np.float32)
    nr, nc = c.shape
    k = min(nr, nc)
    rows, cols = np.empty(k, dtype=np.int32), np.empty(k, dtype=np.int32)
    n = _lib.lib().vkn_lsap_f32(c.ctypes.data, nr, nc, rows.ctypes.data, cols.ctypes.data)
    if n < 0:
        check(n)
    return rows[:n].astype(np.int64), cols[:n].astype(np.int64)

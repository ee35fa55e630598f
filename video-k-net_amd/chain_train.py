"""The [B*N, C] chain of a TRAINING step on the library's own kernels, forward and backward.

Until round 4 the chain of a training step was torch autograd on the BLAS libraries' GEMMs (`KernelUpdateHead._chain_autograd`); the
x-streaming ops on either side of it were already HIP in both directions (`autograd.py`).  Here every layer of the chain is an
`autograd.Function` over the C ABI:

  nn.Linear              forward   bf16x3 split-MFMA GEMM on the weight's tile images (`vkn_linear_f32`); the images of BOTH orientations of
                                   every Linear weight of the chain are rebuilt from the CURRENT values by one launch at the start of the
                                   chain forward (`WeightImages`, `vkn_split_weights_batch_f32` — the weights change every step)
                         backward  dA = dY . W: the same GEMM kernel on the images of the TRANSPOSE;
                                   dW = dY^T . A, db = column sums of dY (exact-fp32 MFMA, deterministic): queued (`DwQueue`) and
                                   computed for the whole chain by ONE launch when its backward is through (`ChainEntryFn`,
                                   `vkn_linear_dw_batch_f32`) — nothing downstream in the chain reads them
  nn.LayerNorm (+ ReLU / sigmoid behind it, + the residual added before it)       `vkn_layernorm_act_{fwd,bwd}_f32`
  the gated update of KernelUpdator.forward between its GEMMs (:70-90)             `UpdatorCoreFn`: `vkn_updator_gate_product_*`, one GEMM
                                                                                   for both gate layers, `vkn_updator_mix_{fwd,bwd}_f32`
  the attention core of nn.MultiheadAttention                                     `vkn_attention_f32` / `vkn_attention_bwd_f32`

and `chain_forward` composes them exactly like `KernelUpdateHead._chain_autograd` (reference: knet/kernel_updator.py:56-93,
knet/det/kernel_update_head.py:198-227, the video links knet/video/kernel_update_head.py:324-476) — the torch chain stays as the A/B
and as the test oracle of this one (`tests/test_gpu_chain_train.py`).  What is left to torch inside the chain: reshapes, the ReLU
mask of the FFN's hidden gradient, autograd's own additions where a tensor feeds two consumers.  No BLAS call, no TunableOp.

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

    def __init__(self, weights, queue=None):
        self.map = {}
        self.queue = queue            # a `DwQueue`: the Linear layers built on these images defer their weight gradients to it
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
        # all N images first, then all T images, each group in list order: the images of consecutive weights are adjacent in memory,
        # which makes the N (T) images of two [C, C] layers the images of their row-wise (column-wise) concatenation (`pair`)
        items, off_n = [], 0
        off_t = sum(_rup(a, 256) for a, _ in sizes)
        for (key, w), (sn, st) in zip(todo, sizes):
            R, Cc = w.shape
            imgn, imgt = buf[off_n:off_n + sn], buf[off_t:off_t + st]
            off_n += _rup(sn, 256)
            off_t += _rup(st, 256)
            items.append(_lib.VknSplitItem(w.data_ptr(), imgn.data_ptr(), Cc, 1, R, Cc, Cc, 0))
            items.append(_lib.VknSplitItem(w.data_ptr(), imgt.data_ptr(), 1, Cc, Cc, _rup(R, 32), R, 0))
            self.map[key] = (imgn, imgt, w)            # (w: keeps a non-contiguous source's copy alive until the launch has run)
        self.buf = buf
        L = _lib.lib()
        with torch.cuda.device(dev):
            for i in range(0, len(items), _lib.SPLIT_MAX_ITEMS):
                chunk = items[i:i + _lib.SPLIT_MAX_ITEMS]
                arr = (_lib.VknSplitItem * len(chunk))(*chunk)
                check(L.vkn_split_weights_batch_f32(arr, len(chunk), _stream()))

    def get(self, w):
        return self.map.get((w.data_ptr(), tuple(w.shape)), (None, None))[:2]

    def pair(self, w1, w2):
        """(N, T) images of the stacked layer [w1; w2] (two [C, C] weights, C % 32 == 0) when this set holds them back to back:
        N = the images of the [2C, C] weight (needs C % 256 == 0: whole 256-row tiles), T = those of its transpose [C, 2C]
        (one row tile, the K tiles of w1^T followed by those of w2^T).  None where that does not hold."""
        n1, t1 = self.get(w1)
        n2, t2 = self.get(w2)
        if n1 is None or n2 is None or w1.shape != w2.shape or w1.shape[0] != w1.shape[1] or w1.shape[0] > 256:
            return None, None
        C = w1.shape[0]
        b0 = self.buf.data_ptr()

        def joined(a, b):
            if a.data_ptr() + a.numel() != b.data_ptr():
                return None
            o = a.data_ptr() - b0
            return self.buf[o:o + a.numel() + b.numel()]

        return (joined(n1, n2) if C % 256 == 0 else None), joined(t1, t2)


def _gemm(a, weight, images, bias, act, K, Nout):
    """act(a [M, K] . Wm^T + bias) with Wm [Nout, K] given by its tile images."""
    M = a.shape[0]
    out = torch.empty((M, Nout), dtype=torch.float32, device=a.device)
    ksplit, ws = 1, None
    if M <= 512 and 256 < K <= 2048 and K % (256 if K < 1024 else 512) == 0 and Nout <= 256:   # (the row epilogue's width)
        # up to 512 rows: the contraction in chunks of 256 (512 from K = 1024 on) over blockIdx.z of the few-row phase kernel
        # (vkn_linear_f32), partial products summed in fixed order by the row epilogue
        ksplit = K // 256 if K < 1024 else K // 512
        ws = torch.empty(ksplit * M * Nout, dtype=torch.float32, device=a.device)
    elif K >= 1024 and Nout <= 256:        # long contraction, few row tiles: split K over workgroups (fixed-order row epilogue)
        ksplit = 8
        ws = torch.empty(ksplit * M * Nout, dtype=torch.float32, device=a.device)
    check(_lib.lib().vkn_linear_f32(_ptr(a), _ptr(weight), _ptr(images), _ptr(bias), _ptr(out), M, K, Nout, int(act), ksplit,
                                    _ptr(ws), ws.numel() * 4 if ws is not None else 0, _stream()))
    return out


class LinearFn(torch.autograd.Function):
    """y = act(a . W^T + b) (wt=False, nn.Linear) or y = a . W + b (wt=True: the folded `feat_transform` weight, used as is).
    img_n / img_t: this weight's tile images from a `WeightImages` (None: built here).  queue: a `DwQueue` — backward then only
    computes da and leaves (dy, a) in the queue; the weight / bias gradients come out of `ChainEntryFn.backward` in one launch."""

    @staticmethod
    def forward(ctx, a, weight, bias, act, wt, img_n, img_t, queue):
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
        ctx.act, ctx.wt, ctx.has_bias, ctx.queue = act, wt, bias is not None, queue
        ctx.save_for_backward(a, weight, y if act else None, img_n if wt else img_t, bias if queue is not None else None)
        return y

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dy):
        a, weight, y, img_b, bias = ctx.saved_tensors
        dy = _f32c(dy, 'dy')
        if ctx.act == 1:
            dy = torch.ops.aten.threshold_backward(dy, y, 0)          # ReLU: one element-wise launch
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
            need_w, need_b = ctx.needs_input_grad[1], ctx.has_bias and ctx.needs_input_grad[2]
            if ctx.queue is not None and not ctx.queue.closed:
                if not (need_w or need_b):
                    return da, None, None, None, None, None, None, None
                # the queue hands a gradient out through the PARAMETER the saved tensor lives in: a tensor it cannot place (a
                # contiguous / aligned copy `_f32c` made of a strided or misaligned parameter, a transposed-weight layer's bias) is
                # computed here instead — never dropped
                if ctx.queue.accepts(weight if need_w else None, bias if need_b else None, ctx.wt):
                    ctx.queue.items.append((dy, a, weight if need_w else None, bias if need_b else None, ctx.wt))
                    return da, None, None, None, None, None, None, None
            if need_w:
                dw = torch.empty_like(weight)
                db = torch.empty(Nout, dtype=torch.float32, device=a.device) if need_b else None
                if ctx.wt:                                 # dW[k][n] = sum_m a[m][k] dy[m][n]
                    check(L.vkn_linear_dw_f32(_ptr(a), K, _ptr(dy), Nout, _ptr(dw), None, M, Nout, K, 0, _stream()))
                    if db is not None:
                        db = dy.sum(0)
                else:                                      # dW[n][k] = sum_m dy[m][n] a[m][k]
                    check(L.vkn_linear_dw_f32(_ptr(dy), Nout, _ptr(a), K, _ptr(dw), _ptr(db), M, K, Nout, 0, _stream()))
            elif need_b:
                db = dy.sum(0)
        return da, dw, db, None, None, None, None, None


class DwQueue:
    """(dy, a, weight view, bias view, wt) of every Linear layer whose backward has run; `flush` computes all their weight / bias
    gradients with one `vkn_linear_dw_batch_f32` launch per 48 layers — the dW GEMMs are off the backward's critical path.

    The gradients leave through the parameters `ChainEntryFn` was given (`params`): a queued tensor must be (a view into) one of them.
    `accepts` is the gate — `LinearFn.backward` computes anything the queue cannot place itself.  A destination that appears twice
    in one flush (tied weights, a module used twice in one chain) is written by the batch launch once and ACCUMULATED into for every
    further use (`vkn_linear_dw_f32(accumulate=1)`); a bias whose weight is frozen gets `dy.sum(0)`."""

    def __init__(self):
        self.items = []
        self.closed = False       # flushed: a Linear backward that still arrives computes its own gradients
        self.params = ()          # set by ChainEntryFn.forward

    def _owner(self, t):
        """index of the parameter the tensor lives in (whole, 4-byte aligned inside it), or None"""
        lo = t.data_ptr()
        hi = lo + t.numel() * t.element_size()
        for i, p in enumerate(self.params):
            plo = p.data_ptr()
            if plo <= lo and hi <= plo + p.numel() * p.element_size() and p.requires_grad and p.is_contiguous() and (lo - plo) % 4 == 0:
                return i
        return None

    def accepts(self, weight, bias, wt):
        if wt and bias is not None:
            return False                               # (the batch kernel has no bias reduction for the untransposed form)
        for t in (weight, bias):
            if t is not None and (not t.is_contiguous() or self._owner(t) is None):
                return False
        return True

    def flush(self, params):
        """-> gradients aligned with `params` (None where no queued layer touched the parameter)"""
        items, self.items, self.closed = self.items, [], True
        self.params = tuple(params)
        bufs, covered = {}, {}

        def slot(view):
            i = self._owner(view)
            if i is None:                              # (`accepts` let it in, so the parameter set changed under us: loud, not silent)
                raise RuntimeError('DwQueue.flush: a queued weight / bias is not part of the parameters of this chain')
            if i not in bufs:
                bufs[i] = torch.empty_like(params[i], memory_format=torch.contiguous_format)
                covered[i] = set()
            off = (view.data_ptr() - params[i].data_ptr()) // 4
            return i, off

        L = _lib.lib()
        by_m, later, seen = {}, [], set()
        for dy, a, w, b, wt in items:
            M, K, Nout = a.shape[0], a.shape[1], dy.shape[1]
            pw = pb = None
            if w is not None:
                i, off = slot(w)
                pw = bufs[i].data_ptr() + off * 4
                covered[i].add((off, w.numel()))
            if b is not None:
                j, offb = slot(b)
                pb = bufs[j].data_ptr() + offb * 4
                covered[j].add((offb, b.numel()))
            if pw is None:                             # frozen weight, trainable bias
                later.append(('bias', bufs[j].view(-1)[offb:offb + b.numel()], dy, pb in seen))
                seen.add(pb)
                continue
            dup = pw in seen or (pb is not None and pb in seen)
            if dup:                                    # second use of a destination in this flush: accumulate after the batch launches
                later.append(('dw', (dy, a, pw, pb, wt, M, K, Nout), pw in seen, pb is not None and pb in seen))
            else:
                it = (_lib.VknDwItem(a.data_ptr(), dy.data_ptr(), pw, None, a.stride(0), dy.stride(0), K, Nout) if wt else
                      _lib.VknDwItem(dy.data_ptr(), a.data_ptr(), pw, pb, dy.stride(0), a.stride(0), Nout, K))
                by_m.setdefault(M, []).append(it)
            seen.add(pw)
            if pb is not None:
                seen.add(pb)
        for i, spans in covered.items():
            if sum(n for _, n in spans) < bufs[i].numel():   # a packed parameter only partly used by the queued layers: the rest of its
                bufs[i].zero_()                              # gradient is zero (enqueued before the launches below, same stream)
        for M, its in by_m.items():
            for j in range(0, len(its), _lib.DW_MAX_ITEMS):
                chunk = its[j:j + _lib.DW_MAX_ITEMS]
                arr = (_lib.VknDwItem * len(chunk))(*chunk)
                check(L.vkn_linear_dw_batch_f32(arr, len(chunk), M, _stream()))
        for ent in later:
            if ent[0] == 'bias':
                _, dst, dy, acc = ent
                dst.add_(dy.sum(0)) if acc else dst.copy_(dy.sum(0))
                continue
            _, (dy, a, pw, pb, wt, M, K, Nout), acc_w, acc_b = ent
            if pb is not None and acc_b != acc_w:      # (a bias shared without its weight, or the reverse: not a shape a module produces)
                raise NotImplementedError('DwQueue: a bias and its weight are shared differently')
            if wt:
                check(L.vkn_linear_dw_f32(a.data_ptr(), a.stride(0), dy.data_ptr(), dy.stride(0), pw, None, M, Nout, K, 1, _stream()))
            else:
                check(L.vkn_linear_dw_f32(dy.data_ptr(), dy.stride(0), a.data_ptr(), a.stride(0), pw, pb, M, K, Nout, 1, _stream()))
        return [bufs.get(i) for i in range(len(params))]


class ChainEntryFn(torch.autograd.Function):
    """Identity on the chain's inputs whose backward — by construction the LAST node of the chain's backward: it needs the gradients of
    all chain inputs — flushes the `DwQueue` and hands out every Linear weight / bias gradient of the chain."""

    @staticmethod
    def forward(ctx, queue, n_in, *args):
        ctx.queue, ctx.n_in, ctx.params = queue, n_in, args[n_in:]
        queue.params = tuple(args[n_in:])        # what `accepts` places queued weights / biases in
        return tuple(a.view_as(a) for a in args[:n_in])

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, *gin):
        with torch.cuda.device(ctx.params[0].device):
            grads = ctx.queue.flush(ctx.params)
        grads = [g if (g is not None and ctx.needs_input_grad[2 + ctx.n_in + i]) else None for i, g in enumerate(grads)]
        return (None, None) + tuple(gin) + tuple(grads)


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


_NORM_NAMES = ('norm_in', 'norm_out', 'input_norm_in', 'input_norm_out')


class UpdatorCoreFn(torch.autograd.Function):
    """`KernelUpdator.forward` between dynamic_layer / input_layer and fc_layer (knet/kernel_updator.py:70-90) as three launches in
    each direction: gate product, ONE GEMM for both gate layers (the stacked weight's images, `WeightImages.pair`), the mix kernel
    (four LayerNorms, two sigmoids, the gated sum).  params, inputs [M, 2C] -> features [M, C]."""

    @staticmethod
    def forward(ctx, params, inputs, wig, big, wug, bug, eps, img_n, img_t, queue, *norms):
        params, inputs = _f32c(params, 'params'), _f32c(inputs, 'inputs')
        M, C = params.shape[0], params.shape[1] // 2
        L = _lib.lib()
        dev = params.device
        G = torch.empty((M, C), dtype=torch.float32, device=dev)
        F_ = torch.empty((M, C), dtype=torch.float32, device=dev)
        stats = torch.empty((M, 8), dtype=torch.float32, device=dev)
        nw = _lib.VknUpdatorNorms(*[t.data_ptr() for t in norms], big.data_ptr() if big is not None else None,
                                  bug.data_ptr() if bug is not None else None)
        with torch.cuda.device(dev):
            check(L.vkn_updator_gate_product_f32(_ptr(params), _ptr(inputs), _ptr(G), M, C, _stream()))
            GT = _gemm(G, wig, img_n, None, 0, C, 2 * C)
            check(L.vkn_updator_mix_fwd_f32(_ptr(GT), _ptr(params), _ptr(inputs), ctypes.byref(nw), float(eps), _ptr(F_), _ptr(stats), M, C,
                                            _stream()))
        ctx.queue = queue
        ctx.save_for_backward(params, inputs, G, GT, stats, wig, big, wug, bug, img_t, *norms)
        return F_

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dF):
        params, inputs, G, GT, stats, wig, big, wug, bug, img_t, *norms = ctx.saved_tensors
        dF = _f32c(dF, 'dF')
        M, C = G.shape
        L = _lib.lib()
        dev = G.device
        dGT, dP, dI = torch.empty_like(GT), torch.empty_like(params), torch.empty_like(inputs)
        dn = [torch.empty_like(t) for t in norms]
        nw = _lib.VknUpdatorNorms(*[t.data_ptr() for t in norms], big.data_ptr() if big is not None else None,
                                  bug.data_ptr() if bug is not None else None)
        gw = _lib.VknUpdatorNormGrads(*[t.data_ptr() for t in dn])
        with torch.cuda.device(dev):
            check(L.vkn_updator_mix_bwd_f32(_ptr(dF), _ptr(GT), _ptr(params), _ptr(inputs), ctypes.byref(nw), _ptr(stats), _ptr(dGT), _ptr(dP),
                                            _ptr(dI), ctypes.byref(gw), M, C, _stream()))
            dG = _gemm(dGT, wig, img_t, None, 0, 2 * C, C)           # [dIG | dUG] . [W_ig ; W_ug]
            check(L.vkn_updator_gate_product_bwd_f32(_ptr(dG), _ptr(params), _ptr(inputs), _ptr(dP), _ptr(dI), M, C, _stream()))
            dwi = dbi = dwu = dbu = None
            if ctx.queue is not None and not ctx.queue.closed:
                ctx.queue.items.append((dGT[:, :C], G, wig, big, False))
                ctx.queue.items.append((dGT[:, C:], G, wug, bug, False))
            else:
                dwi, dwu = torch.empty_like(wig), torch.empty_like(wug)
                dbi = torch.empty_like(big) if big is not None else None
                dbu = torch.empty_like(bug) if bug is not None else None
                check(L.vkn_linear_dw_f32(_ptr(dGT), 2 * C, _ptr(G), C, _ptr(dwi), _ptr(dbi), M, C, C, 0, _stream()))
                check(L.vkn_linear_dw_f32(ctypes.c_void_p(dGT.data_ptr() + 4 * C), 2 * C, _ptr(G), C, _ptr(dwu), _ptr(dbu), M, C, C, 0,
                                          _stream()))
        return (dP, dI, dwi, dbi, dwu, dbu, None, None, None, None) + tuple(dn)


# ---- functional forms
def linear(a, weight, bias=None, act=0, wt=False, images=None):
    img_n, img_t = images.get(weight) if images is not None else (None, None)
    return LinearFn.apply(a, weight, bias, act, wt, img_n, img_t, images.queue if images is not None else None)


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
    img_n, img_t = imgs.pair(ku.input_gate.weight, ku.update_gate.weight) if imgs is not None else (None, None)
    if img_n is not None and img_t is not None and all(getattr(ku, n).elementwise_affine for n in _NORM_NAMES):
        # :70-90 as three launches each way (gate product, one GEMM for both gate layers, the mix kernel)
        norms = [t for n in _NORM_NAMES for t in (getattr(ku, n).weight, getattr(ku, n).bias)]
        features = UpdatorCoreFn.apply(params, inputs, ku.input_gate.weight, ku.input_gate.bias, ku.update_gate.weight, ku.update_gate.bias,
                                       ku.norm_in.eps, img_n, img_t, imgs.queue, *norms)
    else:
        gate_feats = inputs[:, :C] * params[:, :C]                                                    # :70
        input_gate = layernorm(linear(gate_feats, ku.input_gate.weight, ku.input_gate.bias, images=imgs), ku.input_norm_in, act=2)   # :74, :79
        update_gate = layernorm(linear(gate_feats, ku.update_gate.weight, ku.update_gate.bias, images=imgs), ku.norm_in, act=2)      # :75, :80
        param_out = layernorm(params[:, C:], ku.norm_out)                                             # :82
        input_out = layernorm(inputs[:, C:], ku.input_norm_out)                                       # :83
        features = update_gate * param_out + input_gate * input_out                                   # :89-90
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


# ---- the Linear layers a chain uses as (weight, bias | None), in the order it uses them (one `WeightImages` per chain forward)
def _lin(m):
    return (m.weight, m.bias)


def _updator_linears(ku):
    return [_lin(ku.dynamic_layer), _lin(ku.input_layer), _lin(ku.input_gate), _lin(ku.update_gate), _lin(ku.fc_layer)]


def _ffn_linears(ffn):
    return [_lin(m[0]) if isinstance(m, nn.Sequential) else _lin(m) for m in ffn.layers if isinstance(m, (nn.Sequential, nn.Linear))]


def _link_linears(head, names):
    upd, att, _, ffn, _ = (getattr(head, n) if n is not None else None for n in names)
    C = head.in_channels
    w, b = att.attn.in_proj_weight, att.attn.in_proj_bias
    return ((_updator_linears(upd) if upd is not None else []) + [(w[:C], b[:C]), (w[C:], b[C:]), _lin(att.attn.out_proj)]
            + _ffn_linears(ffn))


def chain_linears(head, has_prev):
    C = head.in_channels
    ls = []
    if has_prev and getattr(head, 'previous_link', None) is not None:
        ls += _link_linears(head, head._link_names('link'))
    a = head.attention.attn
    ls += _updator_linears(head.kernel_update_conv) + [(a.in_proj_weight, a.in_proj_bias), _lin(a.out_proj)]
    if head.with_ffn:
        ls += _ffn_linears(head.ffn)
    if has_prev and getattr(head, 'previous', None) is not None and head.previous_type is not None:
        ls += _link_linears(head, head._link_names('track'))
    ls += [_lin(m) for m in list(head.cls_fcs) + list(head.mask_fcs) if isinstance(m, nn.Linear)]
    if getattr(head, 'fc_cls', None) is not None:
        ls.append(_lin(head.fc_cls))
    ls.append(_lin(head.fc_mask))
    if head.feat_transform is not None:
        ls.append((head.feat_transform.conv.weight.reshape(C, C), None))
        ls.append((head.feat_transform.conv.bias.view(1, C), None))
    return ls


def _owners(linears, named_params):
    """The distinct parameters the (views of) weights and biases live in, as `ChainEntryFn` inputs."""
    spans = [(p.data_ptr(), p.data_ptr() + p.numel() * p.element_size(), p) for p in named_params]
    out, seen = [], set()
    for w, b in linears:
        for t in (w, b):
            if t is None:
                continue
            p = next((q for lo, hi, q in spans if lo <= t.data_ptr() < hi), None)
            if p is not None and id(p) not in seen:
                seen.add(id(p))
                out.append(p)
    return out


LN_MAX_WIDTH = 256       # vkn_layernorm_act_*: 64 lanes x LN_MAXV values
ATTN_BWD_MAX_KEYS = 256  # vkn_attention_bwd_f32: AB_THREADS / 2 keys per query


def supported(head, num_kernels=None):
    """Can `chain_forward` run this head's chain on the library's kernels?  Linear in-features % 32 (tile images), LayerNorm widths <=
    256, head width a power of two in 4..64, one eps for the updator's norms, at most 256 kernels per frame in the attention
    backward.  `KernelUpdateHead._chain_impl` asks once per (head, kernel count) and takes the torch autograd chain when this is
    False instead of raising in the middle of a training step."""
    C = head.in_channels
    if head.conv_kernel_size != 1 or C % 32 or C > LN_MAX_WIDTH:
        return False
    try:
        lin = chain_linears(head, hasattr(head, 'attention_previous') and getattr(head, 'attention_previous', None) is not None)
    except Exception:  # noqa: BLE001  (a module tree this function does not know)
        return False
    for w, _ in lin:
        if w.dim() != 2 or w.shape[1] % 32:
            return False
    for m in head.modules():
        if isinstance(m, nn.LayerNorm) and (len(m.normalized_shape) != 1 or m.normalized_shape[0] > LN_MAX_WIDTH):
            return False
    ku = head.kernel_update_conv
    if len({n.eps for n in (ku.norm_in, ku.norm_out, ku.input_norm_in, ku.input_norm_out)}) != 1:
        return False
    hd = C // head.num_heads
    if C % head.num_heads or hd < 4 or hd > 64 or (hd & (hd - 1)):
        return False
    if num_kernels is not None and num_kernels > ATTN_BWD_MAX_KEYS:
        return False
    return True


def chain_forward(head, x_feat, proposal_feat, previous_obj_feats=None):
    """`KernelUpdateHead._chain_autograd` on the library's kernels: same arguments, same five results."""
    B, N = proposal_feat.shape[:2]
    C, K = head.in_channels, head.conv_kernel_size
    M = B * N
    xf = x_feat.reshape(M, C)
    pf = proposal_feat.reshape(M, C)                                                                  # K*K == 1
    prev = previous_obj_feats.reshape(M, C) if previous_obj_feats is not None else None
    linears = chain_linears(head, prev is not None)
    queue = DwQueue() if torch.is_grad_enabled() else None
    imgs = WeightImages([w for w, _ in linears], queue)              # every weight's tile images, both orientations: one launch
    if queue is not None:
        # the entry node: its backward runs when the gradients of ALL chain inputs are there, i.e. last — it computes every queued
        # weight / bias gradient in one launch and returns them as the gradients of the parameters it was given
        owners = _owners(linears, [p for p in head.parameters() if p.requires_grad])
        if owners:
            ins = [xf, pf] + ([prev] if prev is not None else [])
            outs = ChainEntryFn.apply(queue, len(ins), *ins, *owners)
            xf, pf = outs[0], outs[1]
            prev = outs[2] if prev is not None else None
        else:
            imgs.queue = None
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
        kb = linear(mask_feat, ft.bias.view(1, C), images=imgs).view(B, N)         # a one-row layer: mask_feat . b_ft
    else:
        kern, kb = mask_feat, None
    return cls_score, kern.view(B, N, C), kb, obj.reshape(B, N, C, K, K), track

"""GPU (-m gpu): the training chain on the library's own kernels (video-k-net_amd/chain_train.py, csrc/vkn_train.hip) — every building
block forward AND backward against torch fp64 autograd of the same op, then the whole [B*N, C] chain (image head, the "ffn" tracking
link, the previous_link "update" heads) against `KernelUpdateHead._chain_autograd`, the torch restatement of the reference lines
(knet/kernel_updator.py:56-93, knet/det/kernel_update_head.py:198-227, knet/video/kernel_update_head.py:324-476).

Tolerances (written where they are used): forward 2e-5 of the tensor's max-abs (bf16x3 split products, fp32 accumulation — the level
of an fp32 GEMM); gradients 5e-5 of the max-abs for single ops, 5e-4 for the whole chain (a dozen layers deep, fp32 throughout).
"""
import pytest
import torch
import torch.nn.functional as F

from helpers import maxabs
from oracle import synth

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _rand(shape, salt, std=1.0):
    return torch.from_numpy(synth.normalish(shape, salt, std)).to(DEV)


def _rel(got, ref):
    return maxabs(got, ref) / max(float(ref.abs().max()), 1e-30)


@pytest.mark.parametrize('M,K,Nout,act,bias', [(468, 256, 512, 0, True), (468, 2048, 256, 0, True), (468, 256, 2048, 1, True),
                                               (117, 256, 19, 0, True), (33, 64, 64, 0, False), (1000, 256, 768, 0, True),
                                               # 129 - 512 rows: the few-row phase kernel as a plain GEMM (K = 256), in chunks over
                                               # blockIdx.z + the row epilogue (K = 512 / 768 / 2048, at most 256 columns), and what falls
                                               # back (wider outputs of a long contraction; found by tools/soak.py)
                                               (468, 256, 19, 1, True), (300, 512, 256, 0, True), (300, 768, 19, 1, True),
                                               (300, 1024, 300, 0, True), (512, 768, 2048, 1, False), (200, 1536, 64, 0, True), (129, 1280, 256, 0, True)],
                         ids=lambda v: str(v))
def test_linear_forward_backward_vs_fp64(vkn, M, K, Nout, act, bias):
    ct = vkn.chain_train
    a = _rand((M, K), 11).requires_grad_(True)
    w = _rand((Nout, K), 12, 0.05).requires_grad_(True)
    b = _rand((Nout,), 13, 0.1).requires_grad_(True) if bias else None
    gy = _rand((M, Nout), 14, 1e-2)
    y = ct.linear(a, w, b, act=act)
    y.backward(gy)
    ad, wd = a.detach().double().requires_grad_(True), w.detach().double().requires_grad_(True)
    bd = b.detach().double().requires_grad_(True) if bias else None
    yr = F.linear(ad, wd, bd)
    if act:
        yr = torch.relu(yr)
    yr.backward(gy.double())
    assert _rel(y, yr) < 2e-5
    assert _rel(a.grad, ad.grad) < 5e-5
    assert _rel(w.grad, wd.grad) < 5e-5
    if bias:
        assert _rel(b.grad, bd.grad) < 5e-5


def test_linear_with_the_weight_used_untransposed(vkn):
    """y = a . W (the folded feat_transform weight): forward, da = dy . W^T, dW = a^T . dy."""
    ct = vkn.chain_train
    a = _rand((468, 256), 21).requires_grad_(True)
    w = _rand((256, 256), 22, 0.05).requires_grad_(True)
    gy = _rand((468, 256), 23, 1e-2)
    y = ct.linear(a, w, wt=True)
    y.backward(gy)
    ad, wd = a.detach().double().requires_grad_(True), w.detach().double().requires_grad_(True)
    yr = ad @ wd
    yr.backward(gy.double())
    assert _rel(y, yr) < 2e-5 and _rel(a.grad, ad.grad) < 5e-5 and _rel(w.grad, wd.grad) < 5e-5


@pytest.mark.parametrize('M,C,act,resid,sliced', [(468, 256, 0, False, False), (468, 256, 1, True, False), (468, 256, 2, False, True),
                                                  (117, 128, 1, True, True), (5, 64, 0, True, False), (1000, 256, 2, True, False)],
                         ids=lambda v: str(v))
def test_layernorm_act_forward_backward_vs_fp64(vkn, M, C, act, resid, sliced):
    ct = vkn.chain_train
    wide = _rand((M, 2 * C), 31, 2.0).requires_grad_(True)
    x = wide[:, C:] if sliced else wide[:, :C].contiguous()
    r = _rand((M, C), 32).requires_grad_(True) if resid else None
    ln = torch.nn.LayerNorm(C).to(DEV)
    with torch.no_grad():
        ln.weight.copy_(_rand((C,), 33, 0.5) + 1.0)
        ln.bias.copy_(_rand((C,), 34, 0.3))
    gy = _rand((M, C), 35, 1e-2)
    y = ct.layernorm(x, ln, act=act, resid=r)
    y.backward(gy)
    got = (wide.grad.clone(), r.grad.clone() if resid else None, ln.weight.grad.clone(), ln.bias.grad.clone())
    wd = wide.detach().double().requires_grad_(True)
    xd = wd[:, C:] if sliced else wd[:, :C]
    rd = r.detach().double().requires_grad_(True) if resid else None
    gd, bd = ln.weight.detach().double().requires_grad_(True), ln.bias.detach().double().requires_grad_(True)
    z = F.layer_norm(xd + rd if resid else xd, (C,), gd, bd, ln.eps)
    yr = torch.relu(z) if act == 1 else torch.sigmoid(z) if act == 2 else z
    yr.backward(gy.double())
    assert _rel(y, yr) < 1e-5
    assert _rel(got[0], wd.grad) < 5e-5
    if resid:
        assert _rel(got[1], rd.grad) < 5e-5
    assert _rel(got[2], gd.grad) < 5e-5 and _rel(got[3], bd.grad) < 5e-5


def _attn_ref(q, k, v, B, heads):
    Mq, C = q.shape
    hd = C // heads
    qh = q.view(B, -1, heads, hd).transpose(1, 2)
    kh = k.view(B, -1, heads, hd).transpose(1, 2)
    vh = v.view(B, -1, heads, hd).transpose(1, 2)
    p = torch.softmax(qh @ kh.transpose(-1, -2) / hd ** 0.5, -1)
    return (p @ vh).transpose(1, 2).reshape(Mq, C)


@pytest.mark.parametrize('B,N,C,heads', [(4, 117, 256, 8), (1, 166, 256, 8), (2, 216, 128, 8), (3, 20, 64, 4), (1, 256, 256, 8), (2, 70, 256, 4),
                                         (2, 33, 64, 8)],
                         ids=lambda v: str(v))
def test_attention_packed_forward_backward_vs_fp64(vkn, B, N, C, heads):
    """Head widths 8 (VALU backward), 16 / 32 / 64 (matrix-core backward where its LDS footprint fits: N = 166 and 256 at width 32 fall
    back to the VALU kernel), ragged last blocks."""
    ct = vkn.chain_train
    qkv = _rand((B * N, 3 * C), 41, 0.7).requires_grad_(True)
    go = _rand((B * N, C), 42, 1e-2)
    o = ct.attention(qkv, None, B, heads)
    o.backward(go)
    qd = qkv.detach().double().requires_grad_(True)
    orf = _attn_ref(qd[:, :C], qd[:, C:2 * C], qd[:, 2 * C:], B, heads)
    orf.backward(go.double())
    assert _rel(o, orf) < 2e-5
    assert _rel(qkv.grad, qd.grad) < 5e-5


def test_attention_cross_forward_backward_vs_fp64(vkn):
    ct = vkn.chain_train
    B, Nq, Nk, C, heads = 2, 117, 100, 256, 8
    q = _rand((B * Nq, C), 51, 0.7).requires_grad_(True)
    kv = _rand((B * Nk, 2 * C), 52, 0.7).requires_grad_(True)
    go = _rand((B * Nq, C), 53, 1e-2)
    o = ct.attention(q, kv, B, heads)
    o.backward(go)
    qd, kvd = q.detach().double().requires_grad_(True), kv.detach().double().requires_grad_(True)
    orf = _attn_ref(qd, kvd[:, :C], kvd[:, C:], B, heads)
    orf.backward(go.double())
    assert _rel(o, orf) < 2e-5 and _rel(q.grad, qd.grad) < 5e-5 and _rel(kv.grad, kvd.grad) < 5e-5


def _head(vkn, video, over=None, C=256, ffn=2048, N=117):
    cfgd = vkn.configs.roi_head_cfg(video, C=C, heads=8, ffn=ffn, ncls=19, n_thing=8, n_stuff=11, S=1, up=2, nprop=N - 11,
                                    train_cfg=vkn.configs.rcnn_train_cfg(1), mask_over=over)
    head = vkn.build_head(cfgd)
    torch.manual_seed(5)
    head.init_weights()
    stage = head.mask_head[0].to(DEV).train()
    with torch.no_grad():                      # LayerNorm / bias parameters off their init values (ones / zeros hide mistakes)
        for n, p in stage.named_parameters():
            if p.dim() == 1:
                p.add_(_rand(tuple(p.shape), 61 + len(n), 0.2))
    return stage


def _bulk(got, ref):
    """(relative L2 error, median element error relative to the tensor's max-abs)."""
    got, ref = got.detach().double().cpu(), ref.detach().double().cpu()
    d = (got - ref).abs()
    scale = max(float(ref.abs().max()), 1e-30)
    return float(d.norm() / max(float(ref.norm()), 1e-30)), float(d.median()) / scale


@pytest.mark.parametrize('kind', ['image', 'video_ffn', 'video_update', 'video_update_obj'])
def test_device_chain_forward_and_backward_vs_fp64_chain(vkn, kind):
    """The whole chain on the library's kernels against the torch chain (`_chain_autograd`, the restatement of the reference lines) run
    in fp64.  A ReLU whose pre-activation is within fp32 rounding of zero (about one of the 10^6 hidden units of an FFN at this size)
    takes the other branch in fp32 — in this chain exactly as in torch's own fp32 chain (`tools/chain_train_diag.py` prints both
    against fp64) — and moves one row of a weight gradient / one row of an input gradient by a few per cent (and, through the
    attention's keys, many elements by a little).  So gradients are compared in the bulk: relative L2 < 5e-3 (measured without a
    flip: 1e-6 for every element; a wrong formula is an O(1) error); outputs to 5e-5 of the max-abs."""
    import copy
    over = {'video_update': dict(previous_link='update_dynamic_cov', previous_type='update'),
            'video_update_obj': dict(previous_link='link_atten', previous_type='update_obj')}.get(kind)
    stage = _head(vkn, kind != 'image', over)
    B, N, C = 4, 117, 256
    ins = [_rand((B, N, C), 71, 3.0), _rand((B, N, C, 1, 1), 72)]
    if kind != 'image':
        ins.append(_rand((B, N, C, 1, 1), 73))
    results = {}
    for mode in ('device', 'fp64'):
        st = copy.deepcopy(stage).double() if mode == 'fp64' else stage
        st.zero_grad(set_to_none=True)
        args = [(t.double() if mode == 'fp64' else t).clone().requires_grad_(True) for t in ins]
        outs = vkn.chain_train.chain_forward(st, *args) if mode == 'device' else st._chain_autograd(*args)
        loss = 0
        for j, o in enumerate(outs):
            if o is not None:
                w = _rand(tuple(o.shape), 80 + j, 1e-2)
                loss = loss + (o * (w.double() if mode == 'fp64' else w)).sum()
        loss.backward()
        results[mode] = ([o.detach().clone() if o is not None else None for o in outs], [a.grad.clone() for a in args],
                         {n: p.grad.clone() for n, p in st.named_parameters() if p.grad is not None})
    do, dg, dp = results['device']
    to, tg, tp = results['fp64']
    for a, b in zip(do, to):
        assert (a is None) == (b is None)
        if a is not None:
            assert _rel(a, b) < 5e-5
    assert set(dp) == set(tp)
    for name, a, b in [(f'input {i}', a, b) for i, (a, b) in enumerate(zip(dg, tg))] + [(n, dp[n], tp[n]) for n in tp]:
        l2, med = _bulk(a, b)
        assert l2 < 5e-3, (name, l2, med)


def test_dw_queue_never_drops_a_gradient(vkn):
    """the deferred weight-gradient queue must not lose or overwrite anything —
      * a weight used by TWO layers of one chain (tied / re-used module): the batch launch writes, the second use accumulates;
      * a weight that reaches the layer as a NON-CONTIGUOUS view of its parameter (`_f32c` copies it: the queue cannot place the copy):
        that layer computes its own dW in backward;
      * a FROZEN weight with a trainable bias: the bias gradient is dy.sum(0), no NotImplementedError;
      * a parameter only partly covered by queued layers: the rest of its gradient is zero."""
    ct = vkn.chain_train
    M, C = 96, 64
    x = _rand((M, C), 31).requires_grad_(True)
    w1 = _rand((C, C), 32, 0.1).requires_grad_(True)
    b1 = _rand((C,), 33, 0.1).requires_grad_(True)
    wbig = _rand((C, 2 * C), 34, 0.1).requires_grad_(True)        # used through the strided view wbig[:, ::2]
    wf = _rand((C, C), 35, 0.1)                                    # frozen
    bf = _rand((C,), 36, 0.1).requires_grad_(True)
    packed = _rand((2 * C * C,), 37, 0.1).requires_grad_(True)     # one flat parameter, only its first half is a layer's weight
    wp = packed[:C * C].view(C, C)
    gy = _rand((M, C), 38, 1e-2)

    def run(lin, entry):
        xi = entry(x)
        h = lin(xi, w1, b1)
        h = lin(h, w1, b1)                       # tied
        h = lin(h, wbig[:, ::2], None)           # non-contiguous view of a parameter
        h = lin(h, wf, bf)                       # frozen weight, trainable bias
        h = lin(h, wp, None)                     # half of a packed parameter
        return h

    queue = ct.DwQueue()
    imgs = ct.WeightImages([w1, wbig[:, ::2], wf, wp], queue)
    owners = [w1, b1, wbig, bf, packed]
    y = run(lambda a, w, b: ct.linear(a, w, b, images=imgs), lambda t: ct.ChainEntryFn.apply(queue, 1, t, *owners)[0])
    y.backward(gy)
    got = [t.grad.clone() for t in (x, w1, b1, wbig, bf, packed)]
    for t in (x, w1, b1, wbig, bf, packed):
        t.grad = None
    dd = [t.detach().double().requires_grad_(t.requires_grad) for t in (x, w1, b1, wbig, wf, bf, packed)]
    xd, w1d, b1d, wbigd, wfd, bfd, packedd = dd
    h = F.linear(F.linear(xd, w1d, b1d), w1d, b1d)
    h = F.linear(h, wbigd[:, ::2])
    h = F.linear(h, wfd, bfd)
    yr = F.linear(h, packedd[:C * C].view(C, C))
    yr.backward(gy.double())
    assert _rel(y, yr) < 2e-5
    for nm, g, r in zip(('x', 'w1 (tied)', 'b1 (tied)', 'wbig (strided view)', 'bf (frozen weight)', 'packed (half used)'), got,
                        (xd.grad, w1d.grad, b1d.grad, wbigd.grad, bfd.grad, packedd.grad)):
        assert g is not None and _rel(g, r) < 5e-5, nm
    assert float(got[5][C * C:].abs().max()) == 0.0 and float(got[3][:, 1::2].abs().max()) == 0.0

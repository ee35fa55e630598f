"""GPU (-m gpu): SELF-COMPARISON, not parity evidence — the parity pin of the fused tail is
tests/test_gpu_train.py::test_forward_train_vs_reference_golden (+ the cfg3-size golden), which run THROUGH the fused tail by default.

The fused loss tail of the training loop (video-k-net_amd/train_tail.py; include/vkn.h "the loss tail of a training
stage") against the op-by-op path it replaces (per-image sampler -> `get_targets` -> `loss`, reference
knet/det/kernel_iter_head.py:139-231, kernel_update_head.py:279-441): identical loss keys, values to fp32 reduction-order accuracy,
identical assignments, gradients to 1e-5 of each tensor's scale.  The op-by-op path is itself checked against the reference goldens
(test_gpu_train.py::test_forward_train_vs_reference_golden, which now runs THROUGH the fused tail); this file pins the two paths
to each other, the pieces (`vkn_stage_targets`) to `get_targets` bit for bit, and the error word."""
import math

import numpy as np
import pytest
import torch

from helpers import maxabs
from test_gpu_train import _train_case

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _run(vkn, name, fused, steps=1):
    g, case, head, (x, pf, mp, prev), (gt_masks, gt_labels, gt_sem_seg, gt_sem_cls) = _train_case(vkn, name)
    head.fused_tail = fused
    xd = x.to(DEV).requires_grad_(True)
    pfd = pf.to(DEV).requires_grad_(True)
    metas = [dict() for _ in range(case['B'])]
    for _ in range(steps):
        head.zero_grad(set_to_none=True)
        xd.grad = pfd.grad = None
        if case['video']:
            out = head.forward_train_with_previous(xd, pfd, mp.to(DEV), None, metas, gt_masks, gt_labels, gt_sem_seg=gt_sem_seg,
                                                   gt_sem_cls=gt_sem_cls, previous_obj_feats=prev.to(DEV))
            losses = out[0]
        else:
            losses = head.forward_train(xd, pfd, mp.to(DEV), None, metas, gt_masks, gt_labels, gt_sem_seg=gt_sem_seg, gt_sem_cls=gt_sem_cls)
        total = sum(v for k, v in losses.items() if 'loss' in k)
        total.backward()
    grads = {k: p.grad.detach().clone() for k, p in head.named_parameters() if p.grad is not None}
    return {k: v.detach().clone() for k, v in losses.items()}, xd.grad.clone(), pfd.grad.clone(), grads


@pytest.mark.parametrize('name', ['train_tiny', 'train_video', 'train_cfg'])
def test_fused_tail_equals_the_op_by_op_path(vkn, name):
    la, xa, pa, ga = _run(vkn, name, True, steps=2)      # (two steps: per-step state — bank, status word, label cache — is rebuilt)
    lb, xb, pb, gb = _run(vkn, name, False)
    assert sorted(la) == sorted(lb)
    for k in lb:
        assert la[k].shape == lb[k].shape, k
        assert abs(float(la[k]) - float(lb[k])) < 2e-6 * max(1.0, abs(float(lb[k]))), (k, float(la[k]), float(lb[k]))
    assert maxabs(xa, xb) < 1e-5 * float(xb.abs().max())
    assert maxabs(pa, pb) < 1e-5 * float(pb.abs().max())
    assert sorted(ga) == sorted(gb)
    for k in gb:
        assert maxabs(ga[k], gb[k]) < 1e-5 * max(float(gb[k].abs().max()), 1e-12), k


def test_stage_targets_equal_get_targets_bit_for_bit(vkn):
    """`vkn_stage_targets` against `KernelUpdateHead.get_targets` on the sampler's results: labels, label_weights, the positive rows,
    and — through the bank — every row's mask target and weight."""
    from importlib import import_module
    tt = import_module('video_k_net_amd.train_tail')
    g, case, head, (x, pf, mp, prev), (gt_masks, gt_labels, gt_sem_seg, gt_sem_cls) = _train_case(vkn, 'train_cfg')
    B, N = case['B'], case['nprop']
    Hs, Ws = gt_masks[0].shape[-2:]
    mh = head.mask_head[0]
    S = mh.num_stuff_classes
    torch.manual_seed(3)
    masks = torch.randn(B, N + S, Hs, Ws, device=DEV)
    cls = torch.randn(B, N + S, mh.num_classes, device=DEV)
    tail = tt.TailStep.begin(head, torch.device(DEV), gt_masks, gt_labels, gt_sem_seg, gt_sem_cls)
    assert tail is not None
    a = head.mask_assigner[0]
    a.validate_labels(gt_labels, head.num_thing_classes, status=tail.status)
    res = a.assign_batch([masks[i][:N] for i in range(B)], [cls[i][:N, :head.num_thing_classes] for i in range(B)], tail.gt_views, gt_labels)
    # reference route
    samp = [head.mask_sampler[0].sample(res[i], masks[i], gt_masks[i]) for i in range(B)]
    labels, lw, mt, mw = mh.get_targets(samp, gt_masks, gt_labels, head.train_cfg[0], True, gt_sem_seg=gt_sem_seg, gt_sem_cls=gt_sem_cls)
    pos_ref = mh._targets_stash[1]
    # fused route: run the stage and look at what it wrote
    seen = {}
    orig = tt.StageTailFn.apply

    def spy(cls_score, scaled, t, *rest):
        seen['t'] = t
        return orig(cls_score, scaled, t, *rest)
    tt.StageTailFn.apply = spy
    try:
        assert tail.stage_ok(mh, res, cls, masks)
        out = tail.stage_losses(mh, head.train_cfg[0], res, cls, masks)
    finally:
        tt.StageTailFn.apply = orig
    t = seen['t']
    assert torch.equal(t.labels, labels) and torch.equal(t.label_weights, lw) and torch.equal(t.pos_rows, pos_ref)
    assert torch.equal(t.row_weight, mw[:, 0, 0])
    rows = t.tgt_row.long()
    built = torch.where((rows >= 0).view(-1, 1, 1), t.bank[rows.clamp(min=0)], torch.zeros((), device=DEV))
    assert torch.equal(built, mt)
    rk = torch.full_like(t.rowk, -1)
    rk[pos_ref] = torch.arange(pos_ref.numel(), dtype=torch.int32, device=DEV)
    assert torch.equal(t.rowk, rk)
    ref = mh.loss(None, cls, masks, labels, lw, mt, mw)
    for k in ref:
        assert abs(float(out[k]) - float(ref[k])) < 2e-6 * max(1.0, abs(float(ref[k]))), k
    tail.finish()
    torch.cuda.synchronize()
    assert int(tail.status) == 0


def test_out_of_range_stuff_class_is_reported_not_read(vkn):
    """a stuff class outside [num_thing_classes, num_classes) (the reference: an index error inside `labels[...] = ...`) sets the
    status word; every index the kernels use stays in range (the step completes), and the next poll raises."""
    from importlib import import_module
    mha = import_module('video_k_net_amd.mask_hungarian_assigner')
    g, case, head, (x, pf, mp, prev), (gt_masks, gt_labels, gt_sem_seg, gt_sem_cls) = _train_case(vkn, 'train_cfg')
    bad = [c.clone() for c in gt_sem_cls]
    bad[0][0] = head.mask_head[0].num_classes + 5
    mha.FLAGS.poll(wait=True)
    metas = [dict() for _ in range(case['B'])]
    with pytest.raises(IndexError):     # (from the step's own non-blocking poll when the word is already on the host, else from ours)
        losses = head.forward_train(x.to(DEV), pf.to(DEV), mp.to(DEV), None, metas, gt_masks, gt_labels, gt_sem_seg=gt_sem_seg, gt_sem_cls=bad)
        assert all(bool(torch.isfinite(v).all()) for v in losses.values())
        mha.FLAGS.poll(wait=True)
    mha.FLAGS.poll(wait=True)           # nothing is left behind for the next test


def test_repeated_stuff_class_is_reported_and_too_many_stuff_targets_decline_the_fused_tail(vkn):
    """a class listed twice in gt_sem_cls (the reference keeps the last mask and counts the class once) would race on one
    row and over-count the positives: the fused tail reports it through the status word; more stuff targets than stuff kernels cannot
    be distinct -> `TailStep.begin` declines and the op-by-op path (the reference's semantics) runs."""
    from importlib import import_module
    mha = import_module('video_k_net_amd.mask_hungarian_assigner')
    g, case, head, (x, pf, mp, prev), (gt_masks, gt_labels, gt_sem_seg, gt_sem_cls) = _train_case(vkn, 'train_cfg')
    assert gt_sem_cls[0].numel() >= 2
    dup = [c.clone() for c in gt_sem_cls]
    dup[0][1] = dup[0][0]
    mha.FLAGS.poll(wait=True)
    metas = [dict() for _ in range(case['B'])]
    with pytest.raises(IndexError):
        head.forward_train(x.to(DEV), pf.to(DEV), mp.to(DEV), None, metas, gt_masks, gt_labels, gt_sem_seg=gt_sem_seg, gt_sem_cls=dup)
        assert head._last_tail_fused
        mha.FLAGS.poll(wait=True)
    mha.FLAGS.poll(wait=True)
    S = head.mask_head[0].num_stuff_classes
    many_cls = [torch.cat([c, c])[:S + 1] if c.numel() * 2 > S else c for c in gt_sem_cls]
    many_seg = [torch.cat([m, m])[:S + 1] if m.shape[0] * 2 > S else m for m in gt_sem_seg]
    assert any(c.numel() > S for c in many_cls)
    head.forward_train(x.to(DEV), pf.to(DEV), mp.to(DEV), None, metas, gt_masks, gt_labels, gt_sem_seg=many_seg, gt_sem_cls=many_cls)
    assert not head._last_tail_fused
    mha.FLAGS.poll(wait=True)


@pytest.mark.parametrize('name', ['train_video', 'train_cfg', 'train_video_c256'])
def test_lowres_tail_backward_equals_the_upscaled_gradient_plus_adjoint(vkn, name):
    """Round 6 (self-comparison): the fused tail's backward straight into the low-res logits (vkn_mask_losses_bwd_lowres_f32: a thread re-forms
    its S x S block of up-scaled logits, takes the three losses' gradient there and folds it back with the separable adjoint) against the
    form it replaces — the gradient w.r.t. the up-scaled predictions (vkn_mask_losses_bwd_bank_f32) followed by the upsample's adjoint:
    same losses and gradients to fp32 summation order (x2 and x4, with / without the video link; the losses bit for bit once the forward sums also come from the up-scaled tensor).  Both forms meet the
    reference goldens in test_gpu_train.py."""
    outs = []
    for low in (True, False):
        g, case, head, (x, pf, mp, prev), (gt_masks, gt_labels, gt_sem_seg, gt_sem_cls) = _train_case(vkn, name)
        head.lowres_tail = low
        xd, pfd = x.to(DEV).requires_grad_(True), pf.to(DEV).requires_grad_(True)
        metas = [dict() for _ in range(case['B'])]
        if case['video']:
            out = head.forward_train_with_previous(xd, pfd, mp.to(DEV), None, metas, gt_masks, gt_labels, gt_sem_seg=gt_sem_seg,
                                                   gt_sem_cls=gt_sem_cls, previous_obj_feats=prev.to(DEV))
            losses, last_scaled = out[0], out[4]
        else:
            losses, last_scaled = head.forward_train(xd, pfd, mp.to(DEV), None, metas, gt_masks, gt_labels, gt_sem_seg=gt_sem_seg,
                                                     gt_sem_cls=gt_sem_cls), None
        assert head._last_tail_fused
        total = sum(v for k, v in losses.items() if 'loss' in k)
        if last_scaled is not None:      # a consumer of the RETURNED up-scaled tensor still reaches the logits (LazyUpsampleFn)
            assert last_scaled.requires_grad
            total = total + 1e-3 * (last_scaled ** 2).mean()
        total.backward()
        outs.append(({k: float(v.detach()) for k, v in losses.items()}, xd.grad.clone(), pfd.grad.clone(),
                     {k: p.grad.clone() for k, p in head.named_parameters() if p.grad is not None}))
    (la, xa, pa, ga), (lb, xb, pb, gb) = outs
    # (the low-res tail also takes the forward sums from the low-res logits — vkn_mask_losses_fwd_lowres_f32: other partial-sum order)
    assert sorted(la) == sorted(lb) and all(abs(la[k] - lb[k]) <= 2e-6 * max(1.0, abs(lb[k])) for k in lb), (la, lb)
    from importlib import import_module
    TT = import_module('video_k_net_amd.train_tail').TailStep
    try:     # ... and with the forward sums taken from the up-scaled tensor the losses are the same bits as the up-scaled form's
        TT.lowres_forward = False
        g, case, head, (x, pf, mp, prev), (gt_masks, gt_labels, gt_sem_seg, gt_sem_cls) = _train_case(vkn, name)
        metas = [dict() for _ in range(case['B'])]
        if case['video']:
            lc = head.forward_train_with_previous(x.to(DEV).requires_grad_(True), pf.to(DEV).requires_grad_(True), mp.to(DEV), None, metas,
                                                  gt_masks, gt_labels, gt_sem_seg=gt_sem_seg, gt_sem_cls=gt_sem_cls,
                                                  previous_obj_feats=prev.to(DEV))[0]
        else:
            lc = head.forward_train(x.to(DEV).requires_grad_(True), pf.to(DEV).requires_grad_(True), mp.to(DEV), None, metas, gt_masks,
                                    gt_labels, gt_sem_seg=gt_sem_seg, gt_sem_cls=gt_sem_cls)
        assert {k: float(v.detach()) for k, v in lc.items()} == lb
    finally:
        TT.lowres_forward = True
    assert maxabs(xa, xb) < 2e-5 * float(xb.abs().max()) and maxabs(pa, pb) < 2e-5 * float(pb.abs().max())
    assert sorted(ga) == sorted(gb)
    for k in gb:
        assert maxabs(ga[k], gb[k]) < 2e-5 * max(float(gb[k].abs().max()), 1e-12), k


def test_backward_glue_kernels_vs_torch(vkn):
    """the single-launch glue of the gather / decode backward passes against the torch expressions they replace: bit for bit"""
    ops, vag = vkn.ops, vkn.autograd
    g = torch.Generator(device='cpu').manual_seed(5)
    for shape, mag in (((4, 117, 64, 96), 3e-5), ((2, 15, 8, 16), 7.0), ((1, 3, 5, 7), 0.0), ((3, 40, 256), 1e-9)):
        t = (torch.randn(shape, generator=g) * mag).to(DEV)
        s8 = ops.pow2_scale(t)
        # (the torch expression it replaces, `ldexp(1, k)`, is x * pow(2, k) on the device: 4194303.75 for k = 22 — the kernel's scale
        #  IS the power of two)
        ref = torch.tensor(2.0 ** (10 - math.frexp(float(t.abs().max()))[1]), device=DEV)
        assert float(ops.scale_of(s8)) == float(ref) and float(ops.inv_of(s8)) == 1.0 / float(ref), shape
        assert abs(float(vag._pow2_scale(t)) / float(ref) - 1.0) < 1e-6
        if t.dim() == 4:
            got = ops.scale_pad_rows(t, ops.scale_of(s8))
            assert torch.equal(got, vag._scaled_rows(t, ref))
            assert torch.equal(ops.scale_pad_rows(t, None)[:, :shape[1]], t)
            rows = ops.threshold_rows_f16(t, 0.5)
            want = torch.zeros_like(got, dtype=torch.float16)
            want[:, :shape[1]] = (t >= ops.thr_logit(0.5)).half()
            assert torch.equal(rows, want)
        else:
            Np = (shape[1] + 31) // 32 * 32
            assert torch.equal(ops.transpose_pad(t, ops.scale_of(s8), Np), vag._pad_last((t * ref).transpose(1, 2), Np))
            assert torch.equal(ops.transpose_pad(t, None, shape[1]), t.transpose(1, 2).contiguous())
            dkb = torch.randn(shape[0], shape[1], generator=g).to(DEV)
            a, b = ops.unscale_rows(t, dkb, ops.inv_of(s8), shape[1] - 3)
            assert torch.equal(a, t[:, :shape[1] - 3] * (1.0 / ref)) and torch.equal(b, dkb[:, :shape[1] - 3] * (1.0 / ref))
    # a second call on the same stream finds the scratch word re-zeroed
    big, small = torch.full((100000,), 3.0, device=DEV), torch.full((10,), 1e-3, device=DEV)
    assert float(ops.scale_of(ops.pow2_scale(big))) == 256.0 and float(ops.scale_of(ops.pow2_scale(small))) == 2.0 ** 19
    parts = [torch.randn(3, 5, 11, 13, generator=g).to(DEV) for _ in range(6)]
    want = parts[0] + parts[1]
    for p in parts[2:]:
        want = want + p
    assert torch.equal(ops.sum_tensors(list(parts)), want)
    assert torch.equal(ops.sum_tensors(parts[:1]), parts[0])
    many = [torch.randn(1001, generator=g).to(DEV) for _ in range(11)]     # more than one launch's worth of operands
    assert maxabs(ops.sum_tensors(list(many)), torch.stack(many).double().sum(0)) < 1e-5


def test_x_hub_sums_the_feature_map_gradient_once(vkn):
    """`autograd.x_hub`: gather + decode consumers park their dx in the hub (no autograd accumulation), an ordinary consumer's gradient
    joins them, x.grad = the sum — equal to plain autograd's"""
    vag = vkn.autograd
    g = torch.Generator(device='cpu').manual_seed(6)
    B, N, C, H, W = 2, 15, 64, 8, 16
    x0 = torch.randn(B, C, H, W, generator=g).to(DEV)
    k = torch.randn(B, N, C, generator=g).to(DEV).requires_grad_(True)
    m = torch.randn(B, N, H, W, generator=g).to(DEV)
    gz, gx = torch.randn(B, N, H, W, generator=g).to(DEV) * 1e-3, torch.randn(B, N, C, generator=g).to(DEV) * 1e-2

    def run(hub):
        x = x0.clone().requires_grad_(True)
        xs = vag.x_hub(x) if hub else x
        z1, z2 = vag.mask_decode(xs, k), vag.mask_decode(xs, k * 0.5)
        xr, _ = vag.mask_gather(xs, m, 0.5)
        loss = (z1 * gz).sum() + (z2 * gz).sum() + (xr * gx).sum() + (xs * xs).sum() * 1e-3      # the last term: an ordinary consumer
        k.grad = None
        loss.backward()
        return x.grad.clone(), k.grad.clone()
    (xa, ka), (xb, kb) = run(True), run(False)
    assert maxabs(xa, xb) < 1e-6 * float(xb.abs().max()) and torch.equal(ka, kb)


def test_x_hub_partial_backward_leaves_no_stale_parts(vkn):
    """a backward pass that never reaches the hub's node — `torch.autograd.grad(loss, head_params,
    retain_graph=True)` — parks dx contributions in the hub; they must not be added to the NEXT pass over the same graph.  x.grad of
    a full backward after a partial one == x.grad of a fresh full backward, bit for bit; also after a pass that raised midway."""
    vag = vkn.autograd
    g = torch.Generator(device='cpu').manual_seed(8)
    B, N, C, H, W = 2, 15, 64, 8, 16
    x0 = torch.randn(B, C, H, W, generator=g).to(DEV)
    k0 = torch.randn(B, N, C, generator=g).to(DEV)
    m = torch.randn(B, N, H, W, generator=g).to(DEV)
    gz, gx = torch.randn(B, N, H, W, generator=g).to(DEV) * 1e-3, torch.randn(B, N, C, generator=g).to(DEV) * 1e-2

    class Boom(torch.autograd.Function):
        armed = False

        @staticmethod
        def forward(ctx, t):
            return t.view_as(t)

        @staticmethod
        def backward(ctx, gt):
            if Boom.armed:
                raise RuntimeError('boom')
            return gt

    def graph():
        x = x0.clone().requires_grad_(True)
        k = k0.clone().requires_grad_(True)
        xs = vag.x_hub(x)
        z1 = vag.mask_decode(xs, k)
        xr, _ = vag.mask_gather(xs, m, 0.5)
        loss = (z1 * gz).sum() + (xr * gx).sum() + (Boom.apply(xs) * xs).sum() * 1e-3
        return x, k, loss

    x, k, loss = graph()
    loss.backward()
    want = x.grad.clone()
    x, k, loss = graph()
    (gk,) = torch.autograd.grad(loss, [k], retain_graph=True)          # partial pass: the hub's node never runs
    assert x.grad is None and gk is not None
    loss.backward(retain_graph=True)
    assert torch.equal(x.grad, want), 'stale parts of the partial pass were added to the full pass'
    x.grad = None
    Boom.armed = True
    with pytest.raises(RuntimeError, match='boom'):
        loss.backward(retain_graph=True)                                # a pass that dies between a consumer and the hub
    Boom.armed = False
    x.grad = None
    loss.backward()
    assert torch.equal(x.grad, want)


@pytest.mark.parametrize('variant', ['no_stuff_targets', 'pos_weight_stage_weights', 'no_rank_loss', 'one_frame'])
def test_fused_tail_equals_the_op_by_op_path_on_config_variants(vkn, variant):
    """the corners of `get_targets` / `loss` the reference goldens do not visit, fused tail against the op-by-op path of this package:
    a head without stuff kernels (label weights over ALL class columns, :388), pos_weight != 1 with per-stage loss weights != 1,
    no rank loss, a single frame."""
    from oracle import synth
    n_thing, n_stuff = (5, 0) if variant == 'no_stuff_targets' else (2, 3)
    B, C, H, W, up, nprop, S = (1 if variant == 'one_frame' else 3), 64, 8, 16, 2, 12, 2
    over = dict(loss_rank=None) if variant == 'no_rank_loss' else None
    cfgd = vkn.configs.roi_head_cfg(False, C=C, heads=8, ffn=128, ncls=n_thing + n_stuff, n_thing=n_thing, n_stuff=n_stuff, S=S, up=up,
                                    nprop=nprop, train_cfg=vkn.configs.rcnn_train_cfg(S), mask_over=over)
    if variant == 'pos_weight_stage_weights':
        for c in cfgd['train_cfg']:
            c['pos_weight'] = 2.5
        cfgd['stage_loss_weights'] = [1, 0.5]
    tg = synth.train_targets(B, n_thing, n_stuff, H * up, W * up, 91)
    t = lambda key: [torch.from_numpy(e[key]).to(DEV) for e in tg]  # noqa: E731
    gt_masks, gt_labels = t('gt_masks'), t('gt_labels')
    sem = dict(gt_sem_seg=t('gt_sem_seg'), gt_sem_cls=t('gt_sem_cls')) if n_stuff else {}
    N = nprop + n_stuff
    g = torch.Generator(device='cpu').manual_seed(17)
    x0, pf0 = torch.randn(B, C, H, W, generator=g), torch.randn(B, N, C, 1, 1, generator=g)
    mp = (torch.randn(B, N, H, W, generator=g) * 3).to(DEV)
    out = []
    for fused in (True, False):
        torch.manual_seed(0)
        head = vkn.build_head(cfgd)
        head.init_weights()
        head = head.to(DEV).train()
        head.fused_tail = fused
        x, pf = x0.to(DEV).requires_grad_(True), pf0.to(DEV).requires_grad_(True)
        losses = head.forward_train(x, pf, mp, None, [dict() for _ in range(B)], gt_masks, gt_labels, **sem)
        sum(v for k, v in losses.items() if 'loss' in k).backward()
        out.append((losses, x.grad, pf.grad, {k: p.grad for k, p in head.named_parameters() if p.grad is not None}))
    (la, xa, pa, ga), (lb, xb, pb, gb) = out
    assert sorted(la) == sorted(lb) and (('s0_loss_rank' in la) == (variant != 'no_rank_loss'))
    for k in lb:
        assert la[k].shape == lb[k].shape and abs(float(la[k]) - float(lb[k])) < 2e-6 * max(1.0, abs(float(lb[k]))), (k, float(la[k]), float(lb[k]))
    assert maxabs(xa, xb) < 1e-5 * float(xb.abs().max()) and maxabs(pa, pb) < 1e-5 * float(pb.abs().max())
    assert sorted(ga) == sorted(gb)
    for k in gb:
        assert maxabs(ga[k], gb[k]) < 1e-5 * max(float(gb[k].abs().max()), 1e-12), k


def test_deferred_upscaling_is_materialised_when_a_stage_falls_back(vkn):
    """Round 6: inside the low-res tail the x4 logits of a non-final stage are a `KernelIterHead.DeferredScaled` (never computed).  A stage
    whose fused tail declines must still get the real tensor for the op-by-op path: decline stage 1 by force and compare losses and
    gradients with the step in which nothing is deferred (`lowres_tail = False`)."""
    from importlib import import_module
    tt = import_module('video_k_net_amd.train_tail')
    outs = []
    for force in (True, False):
        g, case, head, (x, pf, mp, prev), (gt_masks, gt_labels, gt_sem_seg, gt_sem_cls) = _train_case(vkn, 'train_video_c256')
        head.lowres_tail = force
        calls = {'n': 0, 'deferred': 0}
        orig = tt.TailStep.stage_ok

        def stage_ok(self, mh, res, cls, scaled, _orig=orig):
            calls['n'] += 1
            calls['deferred'] += int(hasattr(scaled, 'materialize'))
            return False if (force and calls['n'] == 2) else _orig(self, mh, res, cls, scaled)
        tt.TailStep.stage_ok = stage_ok
        try:
            xd, pfd = x.to(DEV).requires_grad_(True), pf.to(DEV).requires_grad_(True)
            out = head.forward_train_with_previous(xd, pfd, mp.to(DEV), None, [dict() for _ in range(case['B'])], gt_masks, gt_labels,
                                                   gt_sem_seg=gt_sem_seg, gt_sem_cls=gt_sem_cls, previous_obj_feats=prev.to(DEV))
        finally:
            tt.TailStep.stage_ok = orig
        assert torch.is_tensor(out[4])                       # what forward_train returns is always a tensor
        if force:
            assert calls['deferred'] >= 2 and not head._last_tail_fused
        sum(v for k, v in out[0].items() if 'loss' in k).backward()
        outs.append(({k: float(v.detach()) for k, v in out[0].items()}, xd.grad.clone(), pfd.grad.clone()))
    (la, xa, pa), (lb, xb, pb) = outs
    assert sorted(la) == sorted(lb) and all(abs(la[k] - lb[k]) <= 2e-6 * max(1.0, abs(lb[k])) for k in lb), (la, lb)
    assert maxabs(xa, xb) < 2e-5 * float(xb.abs().max()) and maxabs(pa, pb) < 2e-5 * float(pb.abs().max())


@pytest.mark.parametrize('name', ['train_cfg', 'train_video_c256'])
def test_post_assign_head_low_res_path_equals_the_upscaled_path(vkn, name):
    """`post_assign=True` (a stage is assigned on its OWN predictions, reference knet/det/kernel_iter_head.py:165-171) through the
    round-6 path — assignment costs from the low-res logits, no up-scaled tensor for the non-final stages — against the same head with
    the low-res forms switched off (`lowres_tail = False`, `lowres_costs = False`): identical assignments, losses and gradients to
    fp32 summation order."""
    outs = []
    for low in (True, False):
        g, case, head, (x, pf, mp, prev), (gt_masks, gt_labels, gt_sem_seg, gt_sem_cls) = _train_case(vkn, name)
        head.post_assign = True
        head.lowres_tail = low
        assigned = []
        for a in head.mask_assigner:
            a.lowres_costs = low
            for meth in ('assign_batch', 'assign_batch_lowres'):
                orig = getattr(a, meth)

                def rec(*args, _orig=orig, **kw):
                    rs = _orig(*args, **kw)
                    assigned.extend(r.gt_inds.clone() for r in rs)
                    return rs
                setattr(a, meth, rec)
        xd, pfd = x.to(DEV).requires_grad_(True), pf.to(DEV).requires_grad_(True)
        metas = [dict() for _ in range(case['B'])]
        if case['video']:
            losses = head.forward_train_with_previous(xd, pfd, mp.to(DEV), None, metas, gt_masks, gt_labels, gt_sem_seg=gt_sem_seg,
                                                      gt_sem_cls=gt_sem_cls, previous_obj_feats=prev.to(DEV))[0]
        else:
            losses = head.forward_train(xd, pfd, mp.to(DEV), None, metas, gt_masks, gt_labels, gt_sem_seg=gt_sem_seg, gt_sem_cls=gt_sem_cls)
        assert head._last_tail_fused
        sum(v for k, v in losses.items() if 'loss' in k).backward()
        outs.append((torch.stack(assigned), {k: float(v.detach()) for k, v in losses.items()}, xd.grad.clone(), pfd.grad.clone()))
    (aa, la, xa, pa), (ab, lb, xb, pb) = outs
    assert torch.equal(aa, ab)
    assert sorted(la) == sorted(lb) and all(abs(la[k] - lb[k]) <= 2e-6 * max(1.0, abs(lb[k])) for k in lb), (la, lb)
    assert maxabs(xa, xb) < 2e-5 * float(xb.abs().max()) and maxabs(pa, pb) < 2e-5 * float(pb.abs().max())

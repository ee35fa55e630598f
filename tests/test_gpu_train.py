"""GPU (-m gpu): the TRAINING path — `forward_train` losses, assignments and gradients of the MI355X head against goldens captured
from the reference's own `forward_train` (oracle/gen_golden.py: TRAIN_CASES), and the backward passes of the two x-streaming
HIP ops against fp64 autograd on the device.

Tolerances: losses 1e-4 relative (fp32 reductions over [num_pos, H, W]); assignments bit-exact; gradients 2e-3 of the tensor's
max-abs (the forward's binarised masks are identical to the reference's on these cases, so gradients agree to fp32 accuracy).
"""
import numpy as np
import pytest
import torch

from helpers import load_golden, make_case, maxabs
from oracle import synth

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _rand(shape, salt, std=1.0):
    return torch.from_numpy(synth.normalish(shape, salt, std))


@pytest.mark.parametrize('shape', [(2, 15, 64, 8, 16), (1, 117, 256, 16, 32), (1, 166, 128, 8, 32)], ids=lambda s: 'x'.join(map(str, s)))
def test_gather_decode_backward_vs_fp64_autograd(vkn, shape):
    """dx, dK, dkb of `Z = K x + kb` and dx of `xraw = M x^T` from the HIP kernels (transposed-operand launches) vs torch fp64."""
    B, N, C, H, W = shape
    x = _rand((B, C, H, W), 501).to(DEV).requires_grad_(True)
    k = _rand((B, N, C), 502, 0.3).to(DEV).requires_grad_(True)
    kb = _rand((B, N), 503).to(DEV).requires_grad_(True)
    gz = (_rand((B, N, H, W), 504) * 1e-4).to(DEV)               # small like a real mean-reduced loss gradient
    z = vkn.autograd.mask_decode(x, k, kb)
    z.backward(gz)
    xd, kd, kbd = x.detach().double().requires_grad_(True), k.detach().double().requires_grad_(True), kb.detach().double().requires_grad_(True)
    zr = torch.einsum('bnc,bchw->bnhw', kd, xd) + kbd[:, :, None, None]
    zr.backward(gz.double())
    assert maxabs(z, zr) < 1e-4
    for got, ref in ((x.grad, xd.grad), (k.grad, kd.grad), (kb.grad, kbd.grad)):
        assert maxabs(got, ref) < 2e-5 * float(ref.abs().max())
    # gather: differentiable w.r.t. x only (the binarised mask carries no gradient, knet/det/kernel_update_head.py:191-192)
    x2 = x.detach().clone().requires_grad_(True)
    m = _rand((B, N, H, W), 505, 2.0).to(DEV).requires_grad_(True)
    gx = (_rand((B, N, C), 506) * 1e-3).to(DEV)
    xraw, cnt = vkn.autograd.mask_gather(x2, m, 0.5)
    xraw.backward(gx)
    bits = (m.detach() >= vkn.ops.thr_logit(0.5)).double()
    ref = torch.einsum('bnhw,bnc->bchw', bits, gx.double())
    assert maxabs(x2.grad, ref) < 2e-5 * float(ref.abs().max())
    assert m.grad is None and not cnt.requires_grad


def _train_case(vkn, name):
    g = dict(np.load(f'{__import__("helpers").GOLDEN}/{name}.npz', allow_pickle=False))
    from helpers import CASE_FIELDS
    case = dict(zip(CASE_FIELDS, (int(v) for v in g['case'])))
    over = None
    if 'plink' in g:
        case['plink'], case['ptype'] = str(g['plink']), str(g['ptype'])
        over = dict(previous_link=case['plink'], previous_type=case['ptype'])
    cfgd = vkn.configs.roi_head_cfg(bool(case['video']), C=case['C'], heads=case['heads'], ffn=case['ffn'], ncls=case['ncls'],
                                    n_thing=case['n_thing'], n_stuff=case['n_stuff'], S=case['S'], up=case['up'],
                                    nprop=case['nprop'], train_cfg=vkn.configs.rcnn_train_cfg(case['S']), mask_over=over)
    head = vkn.build_head(cfgd)
    _, sd, x, pf, mp, prev = make_case(case)
    head.load_state_dict(sd, strict=True)
    head = head.to(DEV).train()
    tg = synth.train_targets(case['B'], case['n_thing'], case['n_stuff'], case['H'] * case['up'], case['W'] * case['up'], case['seed'])
    t = lambda key: [torch.from_numpy(e[key]).to(DEV) for e in tg]  # noqa: E731
    return g, case, head, (x, pf, mp, prev), (t('gt_masks'), t('gt_labels'), t('gt_sem_seg'), t('gt_sem_cls'))


def _check_grad(g, tag, got, tol=2e-3, dirty=None, row_len=None):
    """`got` against the golden gradient `tag`: every element within `tol` of the tensor's maximum.

    * goldens with `<tag>_f64` (the reference's own code evaluated in float64, oracle/gen_golden.py `_train_step`): an element may
      instead meet the float64 evaluation — the fp32 reference sits on ReLU kinks now and then (a pre-activation within rounding of 0
      decides a unit's whole gradient: in `train_video_c256` ONE kernel row of grad_pf is 3.8 % of the maximum away from the float64
      evaluation of the same code while every other row agrees to 4e-4; this package's chain AND a torch-autograd chain on its
      forward give the float64 answer) — for at most 1 % of the elements;
    * sampled goldens (4096 elements + the norm): `dirty` = the kernel rows (b * N + n) whose FORWARD result left the parity tolerance
      because a near-threshold mask bit binarised the other way (re-ordering noise of the fp32 logits) — samples on
      those rows (flat index // row_len) are only bounded loosely."""
    got = got.detach().cpu()
    if tag in g:
        ref = torch.from_numpy(g[tag])
        lim = tol * max(float(ref.abs().max()), 1e-12)
        err = (got.float() - ref).abs()
        if tag + '_f64' in g:
            off = err >= lim
            if bool(off.any()):
                err64 = (got.float() - torch.from_numpy(g[tag + '_f64'])).abs()
                assert bool((err64[off] < lim).all()), (tag, float(err[off].max()), float(err64[off].max()), lim)
                assert int(off.sum()) <= 0.01 * off.numel(), (tag, int(off.sum()), off.numel())
        else:
            assert float(err.max()) < lim, tag
    else:
        idx, val = torch.from_numpy(g[tag + '_idx']), torch.from_numpy(g[tag + '_val'])
        lim = tol * max(float(val.abs().max()), 1e-12)
        err = (got.reshape(-1)[idx] - val).abs()
        if dirty is not None:
            # measured on train_video_cfg3 (tools/diag/train_golden_diag.py): 92 of 4096 samples beyond `tol` on 11 kernel rows — 66 on the 7
            # rows whose forward moved (>= 1e-3), 26 on 4 rows a flip in an EARLIER stage touched (their last-stage forward is back inside
            # the frame's 5e-5 noise, the gradient w.r.t. the stage-0 kernels is not): 0.24 - 1.8 % of the maximum
            on_dirty = torch.isin(idx // row_len, dirty)
            off_clean = int((err[~on_dirty] >= lim).sum())
            assert off_clean <= 0.015 * err.numel(), (tag, off_clean, err.numel())
            assert float(err.max()) < 50 * lim, (tag, float(err.max()), lim)       # (a tenth of the maximum)
        else:
            assert float(err.max()) < lim, (tag, float(err.max()), lim)
        assert abs(float(got.double().norm()) - float(g[tag + '_norm'])) < tol * float(g[tag + '_norm']), tag


# forward rows that may leave the parity tolerance at cfg2 / cfg3 size: the CPU oracle against ITSELF with only its fp32 summation order
# changed re-binarises 5-24 near-threshold mask bits per frame and moves up to 24 of 117 kernel rows (profiles/r05_oracle_reorder_noise.json);
# measured for this golden: 6 and 7 rows of 117 (tools/diag/train_golden_diag.py)
CFG3_DIRTY_ROWS_PER_FRAME = 24


def _forward_train_vs_golden(vkn, name, graphs=False, steps=1):
    """One (or `steps`) training step(s) of the head under its DEFAULT policy (fused loss tail, device chain; `graphs`: the chains as
    captured hipGraphs, the first step captures and the later ones replay) against the golden `name`."""
    g, case, head, (x, pf, mp, prev), (gt_masks, gt_labels, gt_sem_seg, gt_sem_cls) = _train_case(vkn, name)
    metas = [dict() for _ in range(case['B'])]
    big_frames = case['H'] * case['W'] >= 128 * 256
    # record the assignments
    assigned = []
    for a in head.mask_assigner:
        orig = a.assign

        def rec(*args, _orig=orig, **kw):
            r = _orig(*args, **kw)
            assigned.append(r.gt_inds.clone())
            return r
        a.assign = rec
        origb = a.assign_batch      # (the training loop assigns a whole batch per stage: one LSAP launch)

        def recb(*args, _orig=origb, **kw):
            rs = _orig(*args, **kw)
            if _orig.__self__.lsap == 'device':        # (the host path goes through `assign`, recorded above)
                assigned.extend(r.gt_inds.clone() for r in rs)
            return rs
        a.assign_batch = recb
        origl = a.assign_batch_lowres     # (... and straight from the low-res logits whenever the up-scaling is the library's own)

        def recl(*args, _orig=origl, **kw):
            rs = _orig(*args, **kw)
            assigned.extend(r.gt_inds.clone() for r in rs)
            return rs
        a.assign_batch_lowres = recl
    if graphs:
        head.enable_chain_graphs()
    for step in range(steps):
        del assigned[:]
        for p in head.parameters():
            p.grad = None
        xd = x.to(DEV).requires_grad_(True)
        pfd = pf.to(DEV).requires_grad_(True)
        track = None
        if case['video']:
            out = head.forward_train_with_previous(xd, pfd, mp.to(DEV), None, metas, gt_masks, gt_labels, gt_sem_seg=gt_sem_seg,
                                                   gt_sem_cls=gt_sem_cls, previous_obj_feats=prev.to(DEV))
            losses, track = out[0], out[5]
        else:
            losses = head.forward_train(xd, pfd, mp.to(DEV), None, metas, gt_masks, gt_labels, gt_sem_seg=gt_sem_seg,
                                        gt_sem_cls=gt_sem_cls)
        assert sorted(losses) == list(g['loss_keys'])
        assert np.array_equal(torch.stack(assigned).cpu().numpy(), g['assigned']), 'Hungarian assignments must be bit-exact'
        for k, ref in zip(g['loss_keys'], g['loss_vals']):
            assert abs(float(losses[k]) - ref) < 1e-4 * max(1.0, abs(ref)), (k, float(losses[k]), ref)
        total = sum(v for k, v in losses.items() if 'loss' in k)
        dirty = None
        if track is not None:
            rowerr = (track.detach().cpu() - torch.from_numpy(g['track'])).abs().reshape(case['B'], case['N'], -1).amax(-1)
            if big_frames:
                # cfg2 / cfg3 size: a few kernel rows per frame see a re-binarised near-threshold mask bit somewhere in the three stages
                assert int((rowerr >= 1e-3).sum(1).max()) <= CFG3_DIRTY_ROWS_PER_FRAME, (rowerr >= 1e-3).sum(1).tolist()
                assert float(rowerr.median()) < 1e-4         # (the attention spreads a flipped row's change over its frame: 4e-5 .. 6e-5 everywhere)
                dirty = torch.nonzero((rowerr >= 1e-3).reshape(-1)).reshape(-1)
            else:
                assert float(rowerr.max()) < 1e-3
            total = total + 0.01 * (track ** 2).sum()
        assert abs(float(total) - float(g['total'])) < 1e-4 * abs(float(g['total']))
        total.backward()
        _check_grad(g, 'grad_x', xd.grad)
        _check_grad(g, 'grad_pf', pfd.grad, dirty=dirty, row_len=case['C'])
        named = dict(head.named_parameters())
        for i, k in enumerate(g['grad_keys']):
            _check_grad(g, f'grad_{i}', named[str(k)].grad)
        # every parameter received a gradient of the reference's norm
        for k, ref in zip(g['all_keys'], g['all_gnorm']):
            p = named[str(k)]
            if ref < 0:
                assert p.grad is None or float(p.grad.abs().max()) == 0.0, k
            else:
                assert p.grad is not None and abs(float(p.grad.double().norm()) - ref) < 5e-3 * max(ref, 1e-6), (k, ref)
    if graphs:
        assert all(len(h._chain_graphs) >= 1 for h in head.mask_head)
        head.enable_chain_graphs(False)
    return head


@pytest.mark.parametrize('name', ['train_tiny', 'train_video', 'train_video_upd', 'train_cfg', 'train_video_c256'])
def test_forward_train_vs_reference_golden(vkn, name):
    """Losses (every `s{stage}_*` key), per-stage assignments and gradients vs the reference's forward_train
    (knet/det/kernel_iter_head.py:139-231, knet/video/kernel_iter_head.py:255-376, knet/det/kernel_update_head.py:279-441); the head
    runs its default policy: the fused loss tail (train_tail.py) and the device chain."""
    head = _forward_train_vs_golden(vkn, name)
    if name != 'train_video_upd':          # (the previous_link heads run the op-by-op tail)
        assert head._last_tail_fused, 'the default policy must have taken the fused loss tail'


CFG3_GOLDEN = 'train_video_cfg3'   # bench.py --train prints this name as its parity witness


@pytest.mark.parametrize('graphs', [False, True], ids=['eager_chain', 'hipgraph_chain'])
def test_forward_train_at_the_benchmarked_cfg3_size_vs_reference_golden(vkn, graphs):
    """the training step AT THE SIZE `bench.py --train` TIMES — video head, C = 256, N = 100 + 17, 128x256 features,
    x4 (512x1024 loss masks: mask_upsample_stride=4 of the shipped KITTI-STEP video config), two frames, ffn link — against the
    reference's own `forward_train_with_previous` run on the CPU (oracle/gen_golden.py `train_video_cfg3`): every loss 1e-4 relative,
    the Hungarian assignments of every stage bit-exact, gradients w.r.t. x / the kernels / a sample of the parameters 2e-3 of their
    maximum (4096 sampled elements + the norm), every parameter's gradient norm.  The tracking output and the kernel-row gradient
    are judged per kernel row: at most CFG3_DIRTY_ROWS_PER_FRAME rows per frame may carry a re-binarised near-threshold mask bit (measured:
    6 and 7 of 117; the reference against itself under a changed summation order: up to 24), every other row meets the tolerances.  `hipgraph_chain`: the policy bench.py runs (chains
    captured as hipGraphs) — capture step and two replays."""
    head = _forward_train_vs_golden(vkn, CFG3_GOLDEN, graphs=graphs, steps=3 if graphs else 1)
    assert head._last_tail_fused


@pytest.mark.parametrize('name', ['train_tiny', 'train_video_upd'])
def test_chain_as_hipgraphs_equals_eager_chain(vkn, name):
    """`enable_chain_graphs()`: every stage's [B*N, C] chain (forward and backward) replayed from captured hipGraphs gives the losses
    and gradients of the eager chain — over several steps (the first captures, the later ones replay into the same static buffers),
    with `zero_grad` between them, and still against the reference golden."""
    g, case, head, (x, pf, mp, prev), (gt_masks, gt_labels, gt_sem_seg, gt_sem_cls) = _train_case(vkn, name)
    metas = [dict() for _ in range(case['B'])]

    def run(h, scale):
        xd = (x * scale).to(DEV).requires_grad_(True)
        pfd = pf.to(DEV).requires_grad_(True)
        for p in h.parameters():
            p.grad = None
        if case['video']:
            out = h.forward_train_with_previous(xd, pfd, mp.to(DEV), None, metas, gt_masks, gt_labels, gt_sem_seg=gt_sem_seg,
                                                gt_sem_cls=gt_sem_cls, previous_obj_feats=prev.to(DEV))
            losses, total = out[0], 0.01 * (out[5] ** 2).sum()
        else:
            losses = h.forward_train(xd, pfd, mp.to(DEV), None, metas, gt_masks, gt_labels, gt_sem_seg=gt_sem_seg, gt_sem_cls=gt_sem_cls)
            total = 0.0
        total = total + sum(v for k, v in losses.items() if 'loss' in k)
        total.backward()
        return ({k: float(v.detach()) for k, v in losses.items()}, xd.grad.clone(), pfd.grad.clone(),
                {k: p.grad.clone() for k, p in h.named_parameters() if p.grad is not None})

    eager = [run(head, sc) for sc in (1.0, 0.9, 1.0)]
    head.enable_chain_graphs()
    graphed = [run(head, sc) for sc in (1.0, 0.9, 1.0)]
    assert all(len(h._chain_graphs) >= 1 for h in head.mask_head)
    for (le, gxe, gpe, pe), (lg, gxg, gpg, pg) in zip(eager, graphed):
        assert le.keys() == lg.keys() and pe.keys() == pg.keys()
        for k in le:
            assert abs(le[k] - lg[k]) <= 1e-6 * max(1.0, abs(le[k])), (k, le[k], lg[k])
        assert maxabs(gxe, gxg) <= 1e-6 * float(gxe.abs().max()) and maxabs(gpe, gpg) <= 1e-6 * float(gpe.abs().max())
        for k in pe:
            assert maxabs(pe[k], pg[k]) <= 1e-6 * max(float(pe[k].abs().max()), 1e-12), k
    for k, ref in zip(g['loss_keys'], g['loss_vals']):           # and the graphed step is still the reference's step
        assert abs(graphed[0][0][str(k)] - ref) < 1e-4 * max(1.0, abs(ref)), k
    head.enable_chain_graphs(False)
    assert all(h._chain_graphs is None for h in head.mask_head)


@pytest.mark.parametrize('mode', ['zero_in_place', 'accumulate'])
def test_chain_graphs_without_reducer_zero_in_place_and_accumulation(vkn, mode):
    """the captured chains' static gradient buffers become `p.grad`; `optimizer.zero_grad(set_to_none=False)` and
    gradient accumulation over two micro-batches (no reducer) must still give the eager chain's gradients."""
    g, case, head, (x, pf, mp, prev), (gt_masks, gt_labels, gt_sem_seg, gt_sem_cls) = _train_case(vkn, 'train_tiny')
    metas = [dict() for _ in range(case['B'])]

    def step(h, scale):
        xd = (x * scale).to(DEV)
        losses = h.forward_train(xd, pf.to(DEV), mp.to(DEV), None, metas, gt_masks, gt_labels, gt_sem_seg=gt_sem_seg,
                                 gt_sem_cls=gt_sem_cls)
        sum(v for k, v in losses.items() if 'loss' in k).backward()

    def run(h):
        for p in h.parameters():
            p.grad = None
        step(h, 1.0)
        if mode == 'zero_in_place':
            for p in h.parameters():
                if p.grad is not None:
                    p.grad.zero_()
        step(h, 0.9)
        return {k: p.grad.clone() for k, p in h.named_parameters() if p.grad is not None}

    eager = run(head)
    head.enable_chain_graphs()
    run(head)                       # captures
    graphed = run(head)             # replays into the same static buffers
    head.enable_chain_graphs(False)
    assert eager.keys() == graphed.keys()
    for k in eager:
        assert maxabs(eager[k], graphed[k]) <= 2e-6 * max(float(eager[k].abs().max()), 1e-12), k


@pytest.mark.parametrize('set_to_none', [False, True])
def test_chain_graphs_deliver_gradients_to_the_bucketed_reducer(vkn, set_to_none):
    """Captured chains hand their parameter gradients over in bulk (no autograd accumulation nodes): through
    `BucketedGradAllReducer.params_ready` they must land in the buckets exactly like eager gradients do, in both zero_grad modes,
    step after step (the delivered tensors alias the graphs' static buffers)."""
    from importlib import import_module
    vdist = import_module('video_k_net_amd.dist')
    g, case, head, (x, pf, mp, prev), (gt_masks, gt_labels, gt_sem_seg, gt_sem_cls) = _train_case(vkn, 'train_video')
    metas = [dict() for _ in range(case['B'])]

    def grads(h, red, scale):
        xd = (x * scale).to(DEV).requires_grad_(True)
        if red is None:
            for p in h.parameters():
                p.grad = None
        else:
            red.zero_grad(set_to_none=set_to_none)
        out = h.forward_train_with_previous(xd, pf.to(DEV), mp.to(DEV), None, metas, gt_masks, gt_labels, gt_sem_seg=gt_sem_seg,
                                            gt_sem_cls=gt_sem_cls, previous_obj_feats=prev.to(DEV))
        (sum(v for k, v in out[0].items() if 'loss' in k) + 0.01 * (out[5] ** 2).sum()).backward()
        if red is not None:
            red.finalize()
        return {k: p.grad.clone() for k, p in h.named_parameters() if p.grad is not None}

    eager = [grads(head, None, sc) for sc in (1.0, 0.9)]
    red = vdist.BucketedGradAllReducer(head)
    head.enable_chain_graphs()
    assert all(h.on_param_grads is not None for h in head.mask_head)
    for e, sc in zip(eager, (1.0, 0.9)):
        got = grads(head, red, sc)
        for k in e:
            assert k in got and maxabs(e[k], got[k]) <= 1e-6 * max(float(e[k].abs().max()), 1e-12), k
        for b in red.buckets:                                     # every delivered gradient now lives in its bucket's flat buffer
            for p, v in zip(b['params'], b['views']):
                assert p.grad is None or p.grad.data_ptr() == v.data_ptr()
    head.enable_chain_graphs(False)


@pytest.mark.parametrize('B,Ns,H,W,K,with_rank', [(2, 17, 16, 24, 7, True), (4, 117, 32, 64, 40, True), (1, 20, 8, 12, 3, False),
                                                 (3, 33, 24, 20, 33, True)])
def test_fused_mask_losses_vs_torch_ops(vkn, B, Ns, H, W, K, with_rank):
    """`vkn_mask_losses_*` (BCE + dice over the positive rows, rank loss over the kernel axis; one backward pass) against the torch op
    sequence of `KernelUpdateHead.loss` on the same tensors: losses 1e-5 relative, gradient 1e-5 of its maximum."""
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(100 + Ns)
    R = B * Ns
    pred = (torch.randn(B, Ns, H, W, generator=g) * 3).to(DEV)
    pos_rows = torch.sort(torch.randperm(R, generator=g)[:K])[0].to(DEV)
    targets = torch.zeros(R, H, W)
    soft = torch.rand(K, H, W, generator=g)
    targets[pos_rows.cpu()] = torch.where(soft > 0.6, soft, torch.zeros(()))          # soft targets with holes (zeros = not covered)
    targets = targets.to(DEV)
    w_mask, w_dice, eps, w_rank = 1.0, 4.0, 1e-3, (0.1 if with_rank else None)

    def torch_losses(p):
        pp, tt = p.reshape(R, H, W)[pos_rows], targets[pos_rows]
        lm = w_mask * F.binary_cross_entropy_with_logits(pp, tt, reduction='mean')
        sp = pp.sigmoid().flatten(1)
        a, b_, c = (sp * tt.flatten(1)).sum(1), (sp * sp).sum(1) + eps, (tt.flatten(1) ** 2).sum(1) + eps
        ld = w_dice * (1 - 2 * a / (b_ + c)).mean()
        lr = p.new_zeros(())
        if with_rank:
            pos = torch.zeros(R, dtype=torch.bool, device=DEV)
            pos[pos_rows] = True
            covered = targets.view(B, Ns, H, W).bool() & pos.view(B, Ns, 1, 1)
            idx = torch.arange(Ns, device=DEV).view(1, Ns, 1, 1)
            top = torch.where(covered, idx, idx.new_full((), -1)).amax(dim=1)
            tgt = torch.where(top >= 0, top, top.new_full((), 255))
            lr = w_rank * F.cross_entropy(p, tgt, reduction='none', ignore_index=255).mean()
        return lm, ld, lr

    pa = pred.clone().requires_grad_(True)
    la = vkn.autograd.mask_losses(pa, targets, pos_rows, w_mask, w_dice, eps, w_rank)
    pb = pred.clone().requires_grad_(True)
    lb = torch_losses(pb)
    for u, v in zip(la, lb):
        assert abs(float(u.detach()) - float(v.detach())) <= 1e-5 * max(1.0, abs(float(v.detach()))), (float(u.detach()), float(v.detach()))
    ws = (0.7, 1.3, 2.1)
    sum(w * l for w, l in zip(ws, la)).backward()
    sum(w * l for w, l in zip(ws, lb)).backward()
    assert maxabs(pa.grad, pb.grad) <= 1e-5 * float(pb.grad.abs().max()), (maxabs(pa.grad, pb.grad), float(pb.grad.abs().max()))


@pytest.mark.parametrize('M,ncls,weighted', [(468, 124, 2), (117, 19, 0), (7, 3, 1), (3744, 124, 1), (50, 1, 2)])
def test_fused_focal_loss_vs_torch_formula(vkn, M, ncls, weighted):
    """`FocalLoss` on CUDA tensors runs ONE HIP pass (vkn_focal_loss_f32) — value and gradient against the torch restatement of
    mmdet's py_sigmoid_focal_loss (losses.py, `fused=False`), with a device-tensor avg_factor, background and weighted rows."""
    g = torch.Generator().manual_seed(900 + M)
    z = (torch.randn(M, ncls, generator=g) * 4).to(DEV)
    labels = torch.randint(0, ncls + 1, (M,), generator=g).to(DEV)          # ncls = background
    # 0: no weight, 1: per row [M], 2: per element [M, ncls] (the head's label_weights)
    w = None if not weighted else (torch.rand((M,) if weighted == 1 else (M, ncls), generator=g) > 0.2).float().to(DEV)
    avg = torch.tensor(17.0, device=DEV)
    loss = vkn.losses.FocalLoss(use_sigmoid=True, gamma=2.0, alpha=0.25, loss_weight=2.0)
    za, zb = z.clone().requires_grad_(True), z.clone().requires_grad_(True)
    la = loss(za, labels, w, avg_factor=avg)
    loss.fused = False
    lb = loss(zb, labels, w, avg_factor=avg)
    (la * 1.7).backward()
    (lb * 1.7).backward()
    assert abs(float(la) - float(lb)) <= 1e-5 * abs(float(lb)), (float(la), float(lb))
    assert maxabs(za.grad, zb.grad) <= 1e-5 * float(zb.grad.abs().max()) + 1e-9


@pytest.mark.parametrize('shape,S', [((2, 5, 8, 12), 2), ((1, 3, 7, 9), 4), ((3, 4, 16, 32), 2), ((1, 2, 5, 3), 3), ((1, 2, 4, 6), 8),
                                     ((2, 3, 5, 7), 1), ((2, 3, 6, 64), 2), ((1, 2, 9, 128), 2), ((1, 1, 5, 320), 2),
                                     ((2, 3, 6, 64), 4), ((1, 2, 9, 128), 4), ((1, 1, 5, 320), 4), ((1, 2, 1, 64), 4), ((1, 8, 128, 256), 4)])
def test_upsample_backward_vs_torch_autograd(vkn, shape, S):
    """The adjoint of the bilinear xS upsample (HIP, gather form, deterministic) against torch's autograd of F.interpolate."""
    import torch.nn.functional as F
    x = _rand(shape, 700 + S).to(DEV)
    g = _rand(shape[:2] + (shape[2] * S, shape[3] * S), 701 + S).to(DEV)
    xa = x.clone().requires_grad_(True)
    ya = vkn.autograd.upsample_bilinear(xa, S)
    ya.backward(g)
    xb = x.clone().requires_grad_(True)
    yb = F.interpolate(xb, scale_factor=S, mode='bilinear', align_corners=False)
    yb.backward(g)
    assert maxabs(ya, yb) < 1e-6 * max(1.0, float(yb.abs().max()))
    assert maxabs(xa.grad, xb.grad) < 1e-5 * float(xb.grad.abs().max())


def test_soft_gt_assignment_vs_reference(vkn):
    """Soft (bilinearly down-sampled) ground-truth masks: DiceCost / MaskCost use the REAL target values."""
    g = dict(np.load(f'{__import__("helpers").GOLDEN}/assign_soft.npz', allow_pickle=False))
    N, G, ncls, H, W, seed = (int(v) for v in g['case'])
    import torch.nn.functional as F
    logits, cls, gt, labels = (torch.from_numpy(a) for a in synth.assign_inputs(N, G, ncls, 2 * H, 2 * W, seed))
    logits = F.interpolate(logits[None], size=(H, W), mode='bilinear', align_corners=False)[0]
    gt = F.interpolate(gt[None], size=(H, W), mode='bilinear', align_corners=False)[0]
    assert float(((gt > 0) & (gt < 1)).float().mean()) > 0.02
    a = vkn.MaskHungarianAssigner(cls_cost=dict(type='FocalLossCost', weight=2.0), dice_cost=dict(type='DiceCost', weight=4.0, pred_act=True),
                                  mask_cost=dict(type='MaskCost', weight=1.0, pred_act=True))
    cost = a.cost_matrix(logits.to(DEV), cls.to(DEV), gt.to(DEV), labels.to(DEV))
    assert maxabs(cost, g['cost']) < 2e-5
    res = a.assign(logits.to(DEV), cls.to(DEV), gt.to(DEV), labels.to(DEV))
    assert np.array_equal(res.gt_inds.cpu().numpy(), g['gt_inds']) and np.array_equal(res.labels.cpu().numpy(), g['labels'])
    with pytest.raises(IndexError):
        a.cost_matrix(logits.to(DEV), cls.to(DEV), gt.to(DEV), torch.full_like(labels, 255).to(DEV))


def test_conv_kernel_head_video_is_the_image_head_over_flattened_clips(vkn):
    """`ConvKernelHeadVideo` (knet_vis/tracker/kernel_head.py, the VIS models' rpn_head): B clips of T frames arrive as B*T frames;
    per-frame metas / masks / (frame, label) rows are flattened and everything else is `ConvKernelHead` — same kernel-init outputs,
    same losses and gradients as the image head on the flattened batch, same state-dict keys."""
    from helpers import GOLDEN, INIT_FIELDS, make_init_case
    g = dict(np.load(f'{GOLDEN}/rpn_train_tiny.npz', allow_pickle=False))
    p = dict(zip(INIT_FIELDS, (int(v) for v in g['case'])))
    assert p['B'] % 2 == 0
    loc, sem, iw, sw, sb = make_init_case(p)
    kw = dict(num_proposals=p['nprop'], in_channels=p['C'], out_channels=p['C'], num_loc_convs=0, num_seg_convs=0,
              localization_fpn=None, conv_kernel_size=1, semantic_fpn=True, num_classes=p['ncls'], use_binary=True,
              proposal_feats_with_obj=True, feat_downsample_stride=2, feat_refine=False, num_thing_classes=p['n_thing'],
              num_stuff_classes=p['ncls'] - p['n_thing'], cat_stuff_mask=True,
              loss_rank=dict(type='CrossEntropyLoss', use_sigmoid=False, loss_weight=0.1),
              loss_seg=dict(type='FocalLoss', use_sigmoid=True, gamma=2.0, alpha=0.25, loss_weight=1.0),
              loss_mask=dict(type='CrossEntropyLoss', use_sigmoid=True, loss_weight=1.0), loss_dice=dict(type='DiceLoss', loss_weight=4.0),
              train_cfg=dict(assigner=dict(type='MaskHungarianAssigner', cls_cost=dict(type='FocalLossCost', weight=2.0),
                                           dice_cost=dict(type='DiceCost', weight=4.0, pred_act=True),
                                           mask_cost=dict(type='MaskCost', weight=1.0, pred_act=True)),
                             sampler=dict(type='MaskPseudoSampler'), pos_weight=1))
    heads = [vkn.build_head(dict(type=t, **kw)) for t in ('ConvKernelHead', 'ConvKernelHeadVideo')]
    assert sorted(heads[0].state_dict()) == sorted(heads[1].state_dict())
    tg = synth.train_targets(p['B'], p['n_thing'], p['ncls'] - p['n_thing'], 2 * p['H'], 2 * p['W'], p['seed'])
    t = lambda key: [torch.from_numpy(e[key]).to(DEV) for e in tg]  # noqa: E731
    T, nclip = 2, p['B'] // 2
    outs = []
    for video, head in enumerate(heads):
        head.load_state_dict({'init_kernels.weight': iw, 'conv_seg.weight': sw, 'conv_seg.bias': sb}, strict=True)
        head = head.to(DEV).train()
        head._upstream_feats = lambda img: img
        locd, semd = loc.to(DEV).requires_grad_(True), sem.to(DEV).requires_grad_(True)
        if video:
            masks, labels = t('gt_masks'), t('gt_labels')
            ref_metas = [[dict() for _ in range(T)] for _ in range(nclip)]
            clip_masks = [[masks[c * T + j] for j in range(T)] for c in range(nclip)]
            clip_labels = [torch.cat([torch.stack([torch.full_like(labels[c * T + j], j), labels[c * T + j]], dim=1) for j in range(T)])
                           for c in range(nclip)]
            res = head.forward_train((locd, semd), [dict()] * nclip, ref_metas, clip_masks, clip_labels, gt_sem_seg=t('gt_sem_seg'),
                                     gt_sem_cls=t('gt_sem_cls'))
            with torch.no_grad():
                inf = head.eval().simple_test_rpn((locd, semd), [dict()] * nclip, ref_metas)
        else:
            res = head.forward_train((locd, semd), [dict() for _ in range(p['B'])], t('gt_masks'), t('gt_labels'),
                                     gt_sem_seg=t('gt_sem_seg'), gt_sem_cls=t('gt_sem_cls'))
            with torch.no_grad():
                inf = head.eval().simple_test_rpn((locd, semd), [dict() for _ in range(p['B'])])
        sum(v for k, v in res[0].items() if 'loss' in k).backward()
        outs.append((res, inf, locd.grad.clone(), head.init_kernels.weight.grad.clone()))
    (ra, ia, ga, wa), (rb, ib, gb, wb) = outs
    assert ra[0].keys() == rb[0].keys()
    for k in ra[0]:
        assert torch.equal(ra[0][k], rb[0][k]), k
    for u, v in zip(ra[1:4], rb[1:4]):
        assert torch.equal(u, v)
    for u, v in zip(ia, ib):
        assert (u is None and v is None) or torch.equal(u, v)
    assert torch.equal(ga, gb) and torch.equal(wa, wb)


@pytest.mark.parametrize('name', ['rpn_train_tiny', 'rpn_train_cfg'])
def test_conv_kernel_head_forward_train_vs_reference_golden(vkn, name):
    """`ConvKernelHead.forward_train` (knet/det/kernel_head.py:267-336) with the shipped rpn losses / train_cfg: losses, Hungarian
    assignments, what it hands to the roi head, and gradients w.r.t. both feature maps and every parameter — against the
    reference's own forward_train + autograd (oracle/gen_golden.py: RPN_TRAIN_CASES).  The two 1x1 convs and the object-feature
    gather run (and back-propagate) through the HIP decode / gather kernels."""
    from helpers import GOLDEN, INIT_FIELDS, make_init_case
    g = dict(np.load(f'{GOLDEN}/{name}.npz', allow_pickle=False))
    p = dict(zip(INIT_FIELDS, (int(v) for v in g['case'])))
    loc, sem, iw, sw, sb = make_init_case(p)
    head = vkn.build_head(dict(
        type='ConvKernelHead', num_proposals=p['nprop'], in_channels=p['C'], out_channels=p['C'], num_loc_convs=0, num_seg_convs=0,
        localization_fpn=None, conv_kernel_size=1, semantic_fpn=True, num_classes=p['ncls'], use_binary=True,
        proposal_feats_with_obj=True, feat_downsample_stride=2, feat_refine=False, num_thing_classes=p['n_thing'],
        num_stuff_classes=p['ncls'] - p['n_thing'], cat_stuff_mask=True,
        loss_rank=dict(type='CrossEntropyLoss', use_sigmoid=False, loss_weight=0.1),
        loss_seg=dict(type='FocalLoss', use_sigmoid=True, gamma=2.0, alpha=0.25, loss_weight=1.0),
        loss_mask=dict(type='CrossEntropyLoss', use_sigmoid=True, loss_weight=1.0), loss_dice=dict(type='DiceLoss', loss_weight=4.0),
        train_cfg=dict(assigner=dict(type='MaskHungarianAssigner', cls_cost=dict(type='FocalLossCost', weight=2.0),
                                     dice_cost=dict(type='DiceCost', weight=4.0, pred_act=True),
                                     mask_cost=dict(type='MaskCost', weight=1.0, pred_act=True)),
                       sampler=dict(type='MaskPseudoSampler'), pos_weight=1)))
    head.load_state_dict({'init_kernels.weight': iw, 'conv_seg.weight': sw, 'conv_seg.bias': sb}, strict=True)
    head = head.to(DEV).train()
    head._upstream_feats = lambda img: img          # the pass-through neck of the golden: (loc_feats, semantic_feats) are the input
    locd, semd = loc.to(DEV).requires_grad_(True), sem.to(DEV).requires_grad_(True)
    tg = synth.train_targets(p['B'], p['n_thing'], p['ncls'] - p['n_thing'], 2 * p['H'], 2 * p['W'], p['seed'])
    t = lambda key: [torch.from_numpy(e[key]).to(DEV) for e in tg]  # noqa: E731
    assigned = []
    orig = head.assigner.assign

    def rec(*args, **kw):
        r = orig(*args, **kw)
        assigned.append(r.gt_inds.clone())
        return r
    head.assigner.assign = rec
    losses, prop, x_feats, masks, cls = head.forward_train((locd, semd), [dict() for _ in range(p['B'])], t('gt_masks'), t('gt_labels'),
                                                           gt_sem_seg=t('gt_sem_seg'), gt_sem_cls=t('gt_sem_cls'))
    assert cls is None and sorted(losses) == list(g['loss_keys'])
    assert np.array_equal(torch.stack(assigned).cpu().numpy(), g['assigned']), 'Hungarian assignments must be bit-exact'
    for k, ref in zip(g['loss_keys'], g['loss_vals']):
        assert abs(float(losses[k]) - ref) < 1e-4 * max(1.0, abs(ref)), (k, float(losses[k]), ref)
    N = p['nprop'] + p['ncls'] - p['n_thing']
    assert tuple(prop.shape) == (p['B'], N, p['C'], 1, 1) and tuple(masks.shape) == (p['B'], N, p['H'], p['W'])
    assert maxabs(prop, g['proposal_feats']) < 1e-3 * (1 + float(np.abs(g['proposal_feats']).max()))
    rs = masks.detach().double().sum(dim=(-1, -2)).cpu().numpy()
    assert np.max(np.abs(rs - g['mask_rowsum'])) < 1e-4 * p['H'] * p['W'] * 8
    total = sum(v for k, v in losses.items() if 'loss' in k) + 1e-3 * (prop ** 2).mean() + 1e-3 * (masks ** 2).mean()
    assert abs(float(total) - float(g['total'])) < 1e-4 * abs(float(g['total']))
    total.backward()
    _check_grad(g, 'grad_loc', locd.grad)
    _check_grad(g, 'grad_sem', semd.grad)
    named = dict(head.named_parameters())
    assert sorted(named) == list(g['grad_keys'])
    for i, k in enumerate(g['grad_keys']):
        _check_grad(g, f'grad_{i}', named[str(k)].grad)


def test_soft_gather_forward_backward_vs_fp64_autograd(vkn):
    """`use_binary=False` (knet/det/kernel_head.py:246-249): xraw = ([sigmoid(z) > 0.5] sigmoid(z)) x^T with gradients to x AND to the
    mask logits, against torch fp64 autograd of the same expression (a reference dead branch in shipped configs; built in round 4)."""
    B, N, C, H, W = 2, 21, 64, 16, 32
    x = _rand((B, C, H, W), 601).to(DEV).requires_grad_(True)
    z = _rand((B, N, H, W), 602, 2.0).to(DEV).requires_grad_(True)
    g = (_rand((B, N, C), 603) * 1e-3).to(DEV)
    out = vkn.autograd.mask_gather_soft(x, z, 0.5)
    out.backward(g)
    xd, zd = x.detach().double().requires_grad_(True), z.detach().double().requires_grad_(True)
    sg = torch.sigmoid(zd)
    ref = torch.einsum('bnhw,bchw->bnc', sg * (z.detach() >= vkn.ops.thr_logit(0.5)).double(), xd)
    ref.backward(g.double())
    assert maxabs(out, ref) < 2e-5 * float(ref.detach().abs().max())
    assert maxabs(x.grad, xd.grad) < 2e-5 * float(xd.grad.abs().max())
    assert maxabs(z.grad, zd.grad) < 2e-5 * float(zd.grad.abs().max())

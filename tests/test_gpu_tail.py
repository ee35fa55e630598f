"""GPU (-m gpu): the fused loss tail of the training loop (video-k-net_amd/train_tail.py; include/vkn.h "the loss tail of a training
stage") against the op-by-op path it replaces (per-image sampler -> `get_targets` -> `loss`, reference
knet/det/kernel_iter_head.py:139-231, kernel_update_head.py:279-441): identical loss keys, values to fp32 reduction-order accuracy,
identical assignments, gradients to 1e-5 of each tensor's scale.  The op-by-op path is itself checked against the reference goldens
(test_gpu_train.py::test_forward_train_vs_reference_golden, which now runs THROUGH the fused tail); this file pins the two paths
to each other, the pieces (`vkn_stage_targets`) to `get_targets` bit for bit, and the error word."""
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

    def spy(cls_score, scaled, t):
        seen['t'] = t
        return orig(cls_score, scaled, t)
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

"""GPU: the train-time assignment costs straight from the LOW-RES logits (vkn_assign_costs_lowres_batch_f32, csrc/vkn_assign_lr.hip).

The reference assigns on `F.interpolate(mask_preds, scale_factor=S, mode='bilinear', align_corners=False)`
(knet/det/kernel_update_head.py:122-130 -> knet/det/kernel_iter_head.py:150-156, 225-226 -> knet/det/mask_hungarian_assigner.py:160-274).
Checked here: the fused kernel against (i) the same formulas in fp64 on the up-scaled logits, (ii) the library's own three-pass form on
the up-scaled tensor, (iii) scipy's assignment on the fp64 costs; batch independence bit for bit; the shape gate.  The reference goldens
(`test_forward_train_*` in tests/test_gpu_train.py) run through this kernel by default — they are the pin against the reference itself."""
import numpy as np
import pytest
import torch

import oracle.synth as synth

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _case(N, Gs, ncls, h, w, S, seed, soft):
    """per image: low-res logits [N + 3, h, w] (the head's rows beyond N are stuff kernels the assignment must not read), class logits,
    ground truth at S h x S w (0/1, or with soft borders as the reference's down-sampled masks have), labels"""
    lows, clss, gts, labs = [], [], [], []
    for b, G in enumerate(Gs):
        lo_up, cl, gt, lab = synth.assign_inputs(N, G, ncls, S * h, S * w, seed + 7 * b)
        low = torch.nn.functional.avg_pool2d(torch.from_numpy(lo_up)[None], S)[0]          # blobs at the low resolution
        low = torch.cat([low * 1.5, torch.from_numpy(synth.normalish((3, h, w), seed + 90 + b, 3.0))])
        g = torch.from_numpy(gt)
        if soft:
            g = torch.nn.functional.avg_pool2d(g[None], 3, 1, 1)[0].contiguous()            # values in [0, 1], soft borders
        lows.append(low.contiguous()), clss.append(torch.from_numpy(cl)), gts.append(g), labs.append(torch.from_numpy(lab))
    return lows, clss, gts, labs


def _want(low, cl, gt, lab, S):
    up = torch.nn.functional.interpolate(low.double()[None], scale_factor=S, mode='bilinear', align_corners=False)[0]
    p = up.sigmoid()
    p1, p2, gd = p.clamp(0.001, 1.0).flatten(1), p.clamp(0.01, 1.0).flatten(1), gt.double().flatten(1)
    dice = -(2 * p1 @ gd.t()) / ((p1 * p1).sum(1, keepdim=True) + 1e-3 + (gd * gd).sum(1)[None] + 1e-3)
    mcost = -(p2 @ gd.t() + (1 - p2) @ (1 - gd).t()) / gd.shape[1]
    pc = cl.double().sigmoid()
    foc = (-(pc + 1e-12).log() * 0.25 * (1 - pc) ** 2 + (1 - pc + 1e-12).log() * 0.75 * pc ** 2)[:, lab]
    return 2.0 * foc + 4.0 * dice + mcost


CASES = [
    # N, per-image G, ncls, h, w, S, soft
    (100, [20, 7, 33, 1], 19, 128, 256, 4, True),      # cfg3 training size: 512 x 1024 ground truth, 4 images, one with two blocks of 32
    (100, [12, 5], 8, 64, 128, 2, False),              # stride 2 (the det configs)
    (150, [9, 70], 5, 8, 32, 4, True),                 # two groups of 128 kernels; 70 ground truths: two passes of two blocks
    (256, [3], 3, 2, 16, 4, False),                    # one tile row
    (37, [40, 2, 6], 4, 12, 48, 2, True),              # ragged N, stride 2, odd tile count
    (100, [9, 14], 19, 48, 156, 4, True),              # KITTI-STEP's 384 x 1248 frames: 156 columns = 9.75 tiles, 48 rows
    (50, [5, 3], 3, 7, 22, 4, False),                  # odd height (a half tile row at the bottom), ragged width
    (64, [11], 3, 5, 12, 2, True),                     # stride 2: 10 x 24 pixels, both edges ragged
]


@pytest.mark.parametrize('N,Gs,ncls,h,w,S,soft', CASES, ids=['cfg3', 'stride2', 'n150_g70', 'n256', 'n37', 'kitti_step', 'odd_h', 'ragged_s2'])
def test_lowres_costs_vs_fp64_and_vs_the_three_pass_form(vkn, N, Gs, ncls, h, w, S, soft):
    lows, clss, gts, labs = _case(N, Gs, ncls, h, w, S, 11, soft)
    dl, dc, dg, dlab = ([t.to(DEV) for t in v] for v in (lows, clss, gts, labs))
    assert vkn.ops.assign_costs_lowres_supported(N, Gs, h, w, S)
    got = vkn.ops.assign_costs_lowres_batch([l[:N] for l in dl], S, dc, dg, dlab)
    ups = [vkn.ops.upsample_bilinear(l[None], S)[0][:N] for l in dl]
    old = vkn.ops.assign_costs_batch(ups, dc, dg, dlab)
    from scipy.optimize import linear_sum_assignment
    a = vkn.MaskHungarianAssigner(cls_cost=dict(type='FocalLossCost', weight=2.0), dice_cost=dict(type='DiceCost', weight=4.0, pred_act=True),
                                  mask_cost=dict(type='MaskCost', weight=1.0, pred_act=True))
    a.validate_labels(dlab, ncls)
    res = a.assign_batch(ups, dc, dg, dlab, lowres=([l[:N] for l in dl], S))
    a.check_status()
    for b in range(len(Gs)):
        want = _want(lows[b][:N], clss[b], gts[b], labs[b], S)
        scale = max(2.0, float(want.abs().max()))
        err = float((got[b].cpu().double() - want).abs().max())
        assert err < 1e-5 * scale, (b, err)                                        # (the focal term is evaluated in fp32)
        assert float((got[b] - old[b]).abs().max()) < 2e-5 * scale
        r0, c0 = linear_sum_assignment(want.numpy())
        inds = np.zeros(N, dtype=np.int64)
        inds[r0] = c0 + 1
        assert np.array_equal(res[b].gt_inds.cpu().numpy(), inds), b


def test_lowres_costs_do_not_depend_on_the_batch(vkn):
    """an image's cost matrix is the same bits alone, first or last in a batch, and run after run (fixed partial-sum order)"""
    N, Gs, ncls, h, w, S = 100, [20, 7, 33], 19, 32, 64, 4
    lows, clss, gts, labs = _case(N, Gs, ncls, h, w, S, 5, True)
    dl, dc, dg, dlab = ([t.to(DEV) for t in v] for v in (lows, clss, gts, labs))
    f = lambda idx: vkn.ops.assign_costs_lowres_batch([dl[i][:N] for i in idx], S, [dc[i] for i in idx], [dg[i] for i in idx],   # noqa: E731
                                                       [dlab[i] for i in idx])
    whole = [c.clone() for c in f([0, 1, 2])]
    again = f([0, 1, 2])
    rev = f([2, 1, 0])
    for b in range(3):
        assert torch.equal(whole[b], again[b]) and torch.equal(whole[b], rev[2 - b]) and torch.equal(whole[b], f([b])[0])


def test_lowres_shape_gate_and_fallback(vkn):
    """S w % 8 != 0, stride 3 or more than 16 images are not taken: `supported` says so, the C entry point returns VKN_E_SHAPE, and the
    assigner computes the same assignment from the up-scaled tensors instead"""
    assert vkn.ops.assign_costs_lowres_supported(100, [5], 16, 24, 4)             # (ragged maps run since the RAGGED form)
    assert not vkn.ops.assign_costs_lowres_supported(100, [5], 16, 32, 3)
    assert not vkn.ops.assign_costs_lowres_supported(100, [5] * 17, 16, 32, 4)
    assert not vkn.ops.assign_costs_lowres_supported(100, [5], 3, 31, 2)         # 62 up-scaled columns: a lane's 8 pixels would straddle the edge
    assert vkn.ops.assign_costs_lowres_supported(100, [5] * 16, 16, 32, 4)
    N, Gs, ncls, h, w, S = 40, [6, 3], 4, 16, 23, 2
    lows, clss, gts, labs = _case(N, Gs, ncls, h, w, S, 3, False)
    dl, dc, dg, dlab = ([t.to(DEV) for t in v] for v in (lows, clss, gts, labs))
    with pytest.raises(ValueError):
        vkn.ops.assign_costs_lowres_batch([l[:N] for l in dl], S, dc, dg, dlab)
    a = vkn.MaskHungarianAssigner(cls_cost=dict(type='FocalLossCost', weight=2.0), dice_cost=dict(type='DiceCost', weight=4.0, pred_act=True),
                                  mask_cost=dict(type='MaskCost', weight=1.0, pred_act=True))
    a.validate_labels(dlab, ncls)
    ups = [vkn.ops.upsample_bilinear(l[None], S)[0][:N] for l in dl]
    res = a.assign_batch(ups, dc, dg, dlab, lowres=([l[:N] for l in dl], S))
    ref = a.assign_batch(ups, dc, dg, dlab)
    for r, q in zip(res, ref):
        assert torch.equal(r.gt_inds, q.gt_inds)

"""GPU: the loss tail's two low-res kernels against the up-scaled forms they replace, on ragged maps (odd heights / widths, one-row and
one-column maps, both strides) — the shapes the goldens do not visit.  SELF-COMPARISON: the reference pin of both forms is
tests/test_gpu_train.py (goldens through the low-res tail by default) and tests/test_gpu_tail.py.
  forward:  vkn_mask_losses_fwd_lowres_f32   vs   vkn_upsample_bilinear_f32 + vkn_mask_losses_fwd_bank_f32
  backward: vkn_mask_losses_bwd_lowres_f32   vs   vkn_mask_losses_bwd_bank_f32 + vkn_upsample_bilinear_bwd_f32
(knet/det/kernel_update_head.py:122-130 up-scaling, :279-349 loss_mask / loss_dice / loss_rank)"""
import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


@pytest.mark.parametrize('B,Ns,h,w,S,K', [(2, 23, 7, 13, 4, 9), (3, 40, 5, 70, 2, 17), (1, 117, 1, 9, 4, 5), (2, 9, 11, 1, 2, 4), (1, 256, 6, 10, 4, 30)],
                         ids=['7x13x4', '5x70x2', 'one_row', 'one_col', 'ns256'])
def test_lowres_forward_and_backward_on_ragged_maps(vkn, B, Ns, h, w, S, K):
    L, ops = vkn._lib.lib(), vkn.ops
    g = torch.Generator().manual_seed(100 * h + w)
    low = (torch.randn(B, Ns, h, w, generator=g) * 3).to(DEV)
    H, W = S * h, S * w
    P = H * W
    bank = (torch.rand(K, H, W, generator=g) > 0.6).float().to(DEV)
    rowk = torch.full((B * Ns,), -1, dtype=torch.int32)
    tgt = torch.zeros(B * Ns, dtype=torch.int32)
    pos = torch.randperm(B * Ns, generator=g)[:K].sort()[0]
    if Ns == 256:
        pos[-1] = B * Ns - 1                      # the last row of a 256-row frame is positive: top = 255 is a real row index
        pos = pos.unique()
        K = int(pos.numel())
        bank = bank[:K].contiguous()
    rowk[pos] = torch.arange(K, dtype=torch.int32)
    tgt[pos] = torch.arange(K, dtype=torch.int32)
    rowk, tgt, posd = rowk.to(DEV), tgt.to(DEV), pos.to(DEV)
    p = lambda t: ctypes.c_void_p(t.data_ptr())   # noqa: E731
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    scaled = ops.upsample_bilinear(low, S)
    # ---- forward
    nch, nbl, ncl = L.vkn_mask_losses_chunks(P), L.vkn_mask_losses_blocks(P), L.vkn_mask_losses_lowres_chunks(h, w)
    rp0, rk0 = torch.zeros(K, nch, 4, device=DEV), torch.zeros(B, nbl, device=DEV)
    lse0, top0 = torch.zeros(B, P, device=DEV), torch.zeros(B, P, dtype=torch.int32, device=DEV)
    rp1, rk1 = torch.zeros(K, ncl, 4, device=DEV), torch.zeros(B, ncl, device=DEV)
    lse1, top1 = torch.full((B, P), float('nan'), device=DEV), torch.full((B, P), -7, dtype=torch.int32, device=DEV)
    if P % 4 == 0:
        assert L.vkn_mask_losses_fwd_bank_f32(p(scaled), p(bank), p(tgt), p(posd), p(rowk), K, B, Ns, P, 1, p(rp0), p(lse0), p(top0), p(rk0), st) == 0
    assert L.vkn_mask_losses_fwd_lowres_f32(p(low), p(bank), p(tgt), p(rowk), K, B, Ns, h, w, S, 1, p(rp1), p(lse1), p(top1), p(rk1), st) == 0
    torch.cuda.synchronize()
    if P % 4 == 0:
        a, b_ = rp0.double().sum(1), rp1.double().sum(1)
        assert float(((a - b_).abs() / a.abs().clamp(min=1.0)).max()) < 2e-6
        assert torch.equal(top0, top1) and float((lse0 - lse1).abs().max()) < 2e-5        # (every pixel was written: no NaN / -7 left)
        assert abs(float(rk0.double().sum()) - float(rk1.double().sum())) < 2e-6 * max(1.0, abs(float(rk0.double().sum())))
    else:      # (the up-scaled form needs P % 4 == 0: compare with torch)
        z = scaled.double().reshape(B, Ns, P)
        assert float((torch.logsumexp(z, 1).float() - lse1).abs().max()) < 2e-5
    # ---- backward
    a_, bc = (torch.rand(K, generator=g) * 50).to(DEV), (torch.rand(K, generator=g) * 50 + 60).to(DEV)
    one = torch.ones(1, device=DEV)
    out_lr = torch.full_like(low, float('nan'))
    assert L.vkn_mask_losses_bwd_lowres_f32(p(low), p(bank), p(tgt), p(rowk), p(a_), p(bc), p(one), p(one), p(one), 1.0, 4.0, 0.1, K, p(lse1),
                                            p(top1), B, Ns, h, w, S, 1, p(out_lr), st) == 0
    if P % 4 == 0:
        gs = torch.empty_like(scaled)
        assert L.vkn_mask_losses_bwd_bank_f32(p(scaled), p(bank), p(tgt), p(rowk), p(a_), p(bc), p(one), p(one), p(one), 1.0, 4.0, 0.1, K, p(lse1),
                                              p(top1), B, Ns, P, 1, p(gs), st) == 0
        ref = ops.upsample_bilinear_bwd(gs, S)
        torch.cuda.synchronize()
        assert float((out_lr - ref).abs().max()) < 2e-6 * float(ref.abs().max())
    else:
        torch.cuda.synchronize()
        assert bool(torch.isfinite(out_lr).all())


def test_lowres_forward_logsumexp_survives_a_huge_spread(vkn):
    """k_ml_fwd_lr sums exp(z - Mb) against ONE reference point per block (the maximum of the block's taps over all rows); logits spread
    over +-600 underflow that sum at pixels far from the maximal tap — those blocks are redone with the online form: the lse must still be
    torch's logsumexp of the up-scaled logits, with no inf / NaN."""
    L, ops = vkn._lib.lib(), vkn.ops
    B, Ns, h, w, S, K = 2, 31, 9, 20, 4, 3
    g = torch.Generator().manual_seed(9)
    low = (torch.randn(B, Ns, h, w, generator=g) * 200).to(DEV)
    H, W = S * h, S * w
    P = H * W
    bank = (torch.rand(K, H, W, generator=g) > 0.5).float().to(DEV)
    rowk = torch.full((B * Ns,), -1, dtype=torch.int32)
    tgt = torch.zeros(B * Ns, dtype=torch.int32)
    rowk[[3, 17, 40]] = torch.arange(K, dtype=torch.int32)
    tgt[[3, 17, 40]] = torch.arange(K, dtype=torch.int32)
    rowk, tgt = rowk.to(DEV), tgt.to(DEV)
    p = lambda t: ctypes.c_void_p(t.data_ptr())   # noqa: E731
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    ncl = L.vkn_mask_losses_lowres_chunks(h, w)
    rp, rk = torch.zeros(K, ncl, 4, device=DEV), torch.zeros(B, ncl, device=DEV)
    lse, top = torch.full((B, P), float('nan'), device=DEV), torch.full((B, P), -7, dtype=torch.int32, device=DEV)
    assert L.vkn_mask_losses_fwd_lowres_f32(p(low), p(bank), p(tgt), p(rowk), K, B, Ns, h, w, S, 1, p(rp), p(lse), p(top), p(rk), st) == 0
    torch.cuda.synchronize()
    want = torch.logsumexp(ops.upsample_bilinear(low, S).double().reshape(B, Ns, P), 1)
    assert bool(torch.isfinite(lse).all())
    assert float((lse.double() - want).abs().max()) < 1e-4 * float(want.abs().max())
    spread = (ops.upsample_bilinear(low, S).reshape(B, Ns, P).amax(1) - low.reshape(B, Ns, -1).amax((1, 2), keepdim=False)[:, None]).min()
    assert float(spread) < -100.0          # (some pixel's best row really lies far below the map's largest tap: the case exists in this input)

"""Differentiable wrappers of the two x-streaming HIP ops (training, BASELINE cfg3; SURVEY.md §8(a) footnote).

The binarised mask carries no gradient (`.float()` of a bool, knet/det/kernel_update_head.py:191-192), so
  * gather  xraw = M x^T           is differentiable w.r.t. x only:   dx  = M^T dxraw            (decode-shaped)
  * decode  Z = K x + kb           w.r.t. both operands:              dK  = dZ x^T, dkb = sum dZ (gather-shaped, real operand)
                                                                      dx  = K^T dZ              (decode-shaped)
i.e. the backward passes are the SAME two kernel shapes with transposed operands: `vkn_mask_decode_f32` with the roles of
channels and kernels swapped, and `vkn_mask_gather_real_f32`.  Gradient operands are scaled by a power of two into the f16
hi/lo split's range first (exact in fp32) and the result is scaled back.  Dynamic range: ONE power-of-two scale per tensor puts
max|g| at ~2^10; an element more than ~2^34 below the tensor's maximum falls under the smallest f16 subnormal of the low half
(2^-24) and contributes zero — irrelevant for loss gradients (their spread inside one tensor is far smaller) but it is a limit.
"""
import math

import torch

from . import ops


def _pow2_scale(t, target=1024.0):
    """Power of two s (device scalar tensor) with max|t| * s in [target / 2, target]; 1 for an all-zero tensor."""
    m = t.detach().abs().amax()
    e = torch.floor(torch.log2(torch.clamp(m, min=1e-30)))
    s = torch.exp2(torch.clamp(math.floor(math.log2(target)) - e, -100.0, 100.0))   # (no host tensor: a tiny H2D copy is a stream sync)
    return torch.where(m > 0, s, torch.ones_like(s))


def _pad_rows(t, mult):
    """[B, R, ...] -> rows padded with zeros to a multiple of `mult`."""
    r = t.shape[1]
    rp = (r + mult - 1) // mult * mult
    if rp == r:
        return t.contiguous()
    pad = t.new_zeros((t.shape[0], rp - r) + tuple(t.shape[2:]))
    return torch.cat([t, pad], dim=1).contiguous()


def _decode_transposed(rows, kern_t):
    """out[b, c, p] = sum_n kern_t[b, c, n] rows[b, n, p]: the decode kernel with `rows` [B, R, H, W] as its feature map
    (R padded to the kernel's 16-channel contraction step) and `kern_t` [B, C, R] as its kernels."""
    rows_p = _pad_rows(rows, 32)
    k = kern_t
    if rows_p.shape[1] != kern_t.shape[2]:
        k = torch.cat([kern_t, kern_t.new_zeros(kern_t.shape[0], kern_t.shape[1], rows_p.shape[1] - kern_t.shape[2])], dim=2)
    return ops.mask_decode(rows_p, k.contiguous())


class MaskGatherFn(torch.autograd.Function):
    """(xraw, cnt) = gather(x, bit(mask_logits)); backward: dx = bit^T dxraw."""

    @staticmethod
    def forward(ctx, x, mask_logits, hard_mask_thr):
        xraw, cnt = ops.mask_gather(x, mask_logits, hard_mask_thr)
        ctx.need_dx = x.requires_grad
        if ctx.need_dx:
            # backward needs nothing but bit(z >= thr): ONE byte per logit is kept for it instead of the fp32 logits
            # (15 MB -> 3.8 MB per frame and stage at cfg2 size)
            ctx.save_for_backward(mask_logits >= ops.thr_logit(hard_mask_thr))
        ctx.mark_non_differentiable(cnt)
        return xraw, cnt

    @staticmethod
    def backward(ctx, dxraw, _dcnt):
        if not ctx.need_dx:
            return None, None, None
        (bits,) = ctx.saved_tensors                                     # [B, N, H, W] bool
        B, N, H, W = bits.shape
        s = _pow2_scale(dxraw)
        kt = (dxraw * s).transpose(1, 2)                                # [B, C, N]
        if (H * W) % 64 == 0:
            # the decode kernel with the bit rows as a HALF-STORAGE feature map (fp16 {0, 1} is exact, its low half is zero: one MFMA
            # per operand pair instead of three, half the bytes of an fp32 bit tensor), rows padded to the 32-row contraction step
            Np = (N + 31) // 32 * 32
            rows = torch.zeros((B, Np, H, W), dtype=torch.float16, device=bits.device)
            rows[:, :N] = bits
            if Np != N:
                kt = torch.cat([kt, kt.new_zeros(B, kt.shape[1], Np - N)], dim=2)
            dx = ops.mask_decode(rows, kt.contiguous()) / s             # [B, C, H, W]
        else:
            dx = _decode_transposed(bits.to(torch.float32), kt) / s
        return dx, None, None


class MaskDecodeFn(torch.autograd.Function):
    """Z = decode(x, K, kb); backward: dK = dZ x^T, dkb = sum_p dZ, dx = K^T dZ."""

    @staticmethod
    def forward(ctx, x, kernels, bias):
        out = ops.mask_decode(x, kernels, bias)
        ctx.save_for_backward(x, kernels)
        ctx.has_bias = bias is not None
        return out

    @staticmethod
    def backward(ctx, dz):
        x, kernels = ctx.saved_tensors
        dz = dz.contiguous()
        s = _pow2_scale(dz)
        dzs = dz * s
        dk = dkb = dx = None
        if ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2]):
            dk, dkb = ops.mask_gather_real(x, dzs)
            dk, dkb = dk / s, dkb / s
        if ctx.needs_input_grad[0]:
            dx = _decode_transposed(dzs, kernels.transpose(1, 2)) / s
        return dx, dk if ctx.needs_input_grad[1] else None, dkb if (ctx.has_bias and ctx.needs_input_grad[2]) else None


def mask_gather(x, mask_logits, hard_mask_thr=0.5):
    return MaskGatherFn.apply(x, mask_logits, hard_mask_thr)


def mask_decode(x, kernels, bias=None):
    return MaskDecodeFn.apply(x, kernels, bias)


class MaskLossesFn(torch.autograd.Function):
    """(loss_mask, loss_dice, loss_rank | 0) of one training stage from the up-scaled mask logits: two HIP passes forward, one
    backward that writes the gradient of all three into one tensor (csrc/vkn_loss.hip).  Values: those of CrossEntropyLoss(
    use_sigmoid=True), DiceLoss(use_sigmoid, activate) and CrossEntropyLoss over the kernel axis with the masked-max rank target
    (`KernelUpdateHead.loss`); the mask targets carry no gradient."""

    @staticmethod
    def forward(ctx, mask_pred, mask_targets, pos_rows, B, w_mask, w_dice, dice_eps, w_rank):
        R, P = mask_pred.shape[0] * mask_pred.shape[1], mask_pred.shape[2] * mask_pred.shape[3]
        pred = mask_pred.reshape(R, P)
        target = mask_targets.reshape(R, P)
        K = int(pos_rows.shape[0])
        dev = pred.device
        rowk = torch.full((R,), -1, dtype=torch.int32, device=dev)
        rowk[pos_rows] = torch.arange(K, dtype=torch.int32, device=dev)
        with_rank = w_rank is not None
        stats, lse, top, rank_sum = ops.mask_losses_fwd(pred, target, pos_rows, rowk, B, with_rank)
        bce, a, b, c = stats.unbind(1)
        bc = (b + dice_eps) + (c + dice_eps)
        loss_mask = w_mask * (bce.sum() / (K * P))
        loss_dice = w_dice * (1.0 - (2.0 * a) / bc).mean()
        loss_rank = (w_rank * (rank_sum / (B * P))) if with_rank else pred.new_zeros(())
        ctx.save_for_backward(pred, target, rowk, a, bc, lse if with_rank else pred.new_empty(0), top if with_rank else rowk.new_empty(0))
        ctx.meta = (B, K, P, w_mask, w_dice, w_rank, tuple(mask_pred.shape))
        return loss_mask, loss_dice, loss_rank

    @staticmethod
    def backward(ctx, g_mask, g_dice, g_rank):
        pred, target, rowk, a, bc, lse, top = ctx.saved_tensors
        B, K, P, w_mask, w_dice, w_rank, shape = ctx.meta
        with_rank = w_rank is not None
        gd = g_dice * (w_dice / K)
        rowcoef = torch.stack([gd * (-2.0 / bc), gd * (4.0 * a / (bc * bc))], dim=1).contiguous()
        coef = torch.stack([g_mask * (w_mask / (K * P)), (g_rank * (w_rank / (B * P))) if with_rank else g_mask * 0.0]).float().contiguous()
        grad = ops.mask_losses_bwd(pred, target, rowk, rowcoef.float(), coef, lse, top, B, with_rank)
        return grad.reshape(shape), None, None, None, None, None, None, None


def mask_losses(mask_pred, mask_targets, pos_rows, w_mask, w_dice, dice_eps, w_rank=None):
    """mask_pred [B, Ns, H, W] logits, mask_targets [B*Ns, H, W], pos_rows int64 [K > 0] ascending -> (loss_mask, loss_dice, loss_rank)."""
    return MaskLossesFn.apply(mask_pred.contiguous(), mask_targets.contiguous(), pos_rows, mask_pred.shape[0], w_mask, w_dice, dice_eps, w_rank)


class UpsampleBilinearFn(torch.autograd.Function):
    """`F.interpolate(x, scale_factor=S, mode='bilinear', align_corners=False)` on the HIP kernels, forward and adjoint."""

    @staticmethod
    def forward(ctx, masks, scale):
        ctx.scale = int(scale)
        return ops.upsample_bilinear(masks, ctx.scale)

    @staticmethod
    def backward(ctx, grad_out):
        return ops.upsample_bilinear_bwd(grad_out.contiguous(), ctx.scale), None


def upsample_bilinear(masks, scale):
    return UpsampleBilinearFn.apply(masks, scale)


class FocalLossFn(torch.autograd.Function):
    """`loss_weight * sum(focal(z, labels) * row_weight) / avg_factor` with the element losses and their derivative from ONE HIP
    pass (vkn_focal_loss_f32) instead of ~20 element-wise launches forward and ~30 backward."""

    @staticmethod
    def forward(ctx, logits, labels, row_weight, scale, alpha, gamma):
        total, grad = ops.focal_loss_fwd(logits, labels, row_weight, alpha, gamma)
        ctx.save_for_backward(grad, scale)
        return total * scale

    @staticmethod
    def backward(ctx, g):
        grad, scale = ctx.saved_tensors
        return grad * (g * scale), None, None, None, None, None


def focal_loss(logits, labels, row_weight, loss_weight, avg_factor, alpha, gamma):
    """mmdet `FocalLoss(use_sigmoid=True, reduction='mean')` with an `avg_factor`: sum / avg_factor * loss_weight.  `avg_factor` may be
    a device tensor (no synchronisation)."""
    if torch.is_tensor(avg_factor):
        scale = (loss_weight / avg_factor.detach().to(device=logits.device, dtype=torch.float32)).reshape(())
    else:
        scale = torch.full((), float(loss_weight) / float(avg_factor), dtype=torch.float32, device=logits.device)
    return FocalLossFn.apply(logits, labels, row_weight, scale, alpha, gamma)

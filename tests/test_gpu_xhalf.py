"""x stored as fp16 / bf16 (VKN_FLAG_X_F16 / VKN_FLAG_X_BF16, include/vkn.h): the head computes in fp32 either way.

Pin: on x' = float(half(x)) the fp32 kernels must return the SAME BITS as the half-storage kernels on half(x) — the half kernels are
the fp32 ones with the (all-zero) x_lo terms removed.  bf16 goes through an f16 conversion that is exact for 2^-14 <= |x| < 65504;
inputs here respect that range for the bit-exact checks, and a second check with tiny values states the tolerance.
Against the fp32 reference on the unrounded x the deviation is the rounding of x itself; the stated tolerances are checked too."""
import numpy as np
import pytest
import torch

from helpers import load_golden

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _rand(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(shape, generator=g) * scale


def _clamp_tiny(x):
    """keep |x| >= 2^-13 so that bf16 -> f16 is exact (normal f16 range)"""
    s = torch.where(x >= 0, torch.ones_like(x), -torch.ones_like(x))
    return torch.where(x.abs() < 2.0 ** -13, s * 2.0 ** -13, x)


CASES = [(2, 117, 256, 64, 128), (1, 100, 256, 48, 80), (3, 40, 64, 32, 64), (1, 216, 128, 32, 64)]


@pytest.mark.parametrize('dt', [torch.float16, torch.bfloat16])
@pytest.mark.parametrize('B,N,C,H,W', CASES)
def test_half_x_kernels_bit_identical_to_fp32_on_rounded_x(vkn, dt, B, N, C, H, W):
    if N > 128 and C != 128:
        pytest.skip('shape list')
    x = _clamp_tiny(_rand((B, C, H, W), 11)).to(DEV)
    xh = x.to(dt)
    xr = xh.float()
    masks = _rand((B, N, H, W), 12, 4.0).to(DEV)
    kern = _rand((B, N, C), 13, 0.2).to(DEV)
    kb = _rand((B, N), 14).to(DEV)
    hi, lo = vkn.ops.split_planes(kern)
    # gather on logits
    a = vkn.ops.mask_gather(xh, masks)
    b = vkn.ops.mask_gather(xr, masks)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    # decode
    a = vkn.ops.mask_decode_planes(xh, hi, lo, N, kb)
    b = vkn.ops.mask_decode_planes(xr, hi, lo, N, kb)
    assert torch.equal(a, b)
    a = vkn.ops.mask_decode(xh, kern, kb)
    assert torch.equal(a, b)
    # fused decode -> gather
    if vkn._lib.lib().vkn_decode_gather_supported(C, H * W):
        a = vkn.ops.decode_gather(xh, hi, lo, N, kb)
        b = vkn.ops.decode_gather(xr, hi, lo, N, kb)
        assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])


def test_bf16_subnormal_range_tolerance(vkn):
    """bf16 values below the normal f16 range lose bits in the f16 conversion: bounded by 2^-25 per element (absolute)."""
    B, N, C, H, W = 1, 64, 256, 32, 64
    x = (_rand((B, C, H, W), 21) * 1e-5).to(DEV)
    xh = x.to(torch.bfloat16)
    kern = _rand((B, N, C), 22).to(DEV)
    hi, lo = vkn.ops.split_planes(kern)
    a = vkn.ops.mask_decode_planes(xh, hi, lo, N)
    b = vkn.ops.mask_decode_planes(xh.float(), hi, lo, N)
    assert (a - b).abs().max().item() <= C * 2.0 ** -25 * kern.abs().max().item() * 4


@pytest.mark.parametrize('dt,tol', [(torch.float16, 5e-3), (torch.bfloat16, 4e-2)])
def test_half_x_head_matches_fp32_head_on_rounded_x_and_reference_within_tolerance(vkn, dt, tol):
    """Whole fused head (S stages + link + upsample): bit-identical to the fp32 head run on the rounded x; and against the fp32 head
    on the unrounded x, first-stage mask logits within the stated tolerance of the logit scale (later stages compound through the
    hard threshold like any perturbation — the chaos note in DESIGN.md — and are compared through their stable rows elsewhere)."""
    from test_gpu_parity import _build_head
    _, case = load_golden('video_cfg')
    head, _ = _build_head(vkn, case)
    T, N, C, H, W = 3, case['N'], case['C'], case['H'], case['W']
    xs = _clamp_tiny(_rand((T, C, H, W), 31)).to(DEV)
    pfs = _rand((T, N, C), 32).to(DEV)
    mps = _rand((T, N, H, W), 33, 4.0).to(DEV)
    first = _rand((1, N, C), 34).to(DEV)
    dims = head.mask_head[0].make_dims(T, N, H, W)
    packs = [h.stage_pack(torch.device(DEV)) for h in head.mask_head]
    xh = xs.to(dt)
    a = vkn.ops.head_forward(dims, packs, xh, pfs, mps, None, case['up'], clip_first_prev=first)
    b = vkn.ops.head_forward(dims, packs, xh.float(), pfs, mps, None, case['up'], clip_first_prev=first)
    for u, v in zip(a, b):
        assert (u is None and v is None) or torch.equal(u, v)
    # the two-kernel hand-off (bit words) takes the half x as well, same bits
    c = vkn.ops.head_forward(dims, packs, xh, pfs, mps, None, case['up'], clip_first_prev=first, flags=vkn.ops.FLAG_BITS_HANDOFF)
    for u, v in zip(a, c):
        assert (u is None and v is None) or torch.equal(u, v)
    # one stage against the unrounded fp32 x
    one = [packs[0]]
    ra = vkn.ops.head_forward(dims, one, xh, pfs, mps, None, 1)
    rb = vkn.ops.head_forward(dims, one, xs, pfs, mps, None, 1)
    scale = rb[2].abs().max().item()
    assert (ra[2] - rb[2]).abs().max().item() <= tol * scale, ((ra[2] - rb[2]).abs().max().item(), scale)


@pytest.mark.parametrize('golden,dt,tol,H,W,T', [('video_cfg', torch.bfloat16, 4e-2, 128, 256, 2),          # BASELINE cfg2 "bf16": 1024x2048, N = 117
                                                 ('video_vipseg_big', torch.float16, 5e-3, 92, 160, 2)],   # BASELINE cfg5 "fp16": 720p, N = 166
                         ids=['cfg2_bf16_128x256_N117', 'cfg5_fp16_92x160_N166'])
def test_half_x_head_at_the_sizes_that_name_it(vkn, golden, dt, tol, H, W, T):
    """The two BASELINE configs that WORD a half type, at their own feature size and kernel count: the whole fused head (3 stages,
    clip link, x4 upsample) on half-storage x is bit-identical to the fp32 head on the rounded x (all three stage hand-offs), and
    the first stage's logits stay within the stated tolerance of the logit scale against the unrounded fp32 x."""
    from test_gpu_parity import _build_head
    _, case = load_golden(golden)
    head, _ = _build_head(vkn, case)
    N, C = case['N'], case['C']
    xs = _clamp_tiny(_rand((T, C, H, W), 61)).to(DEV)
    pfs = _rand((T, N, C), 62).to(DEV)
    mps = _rand((T, N, H, W), 63, 4.0).to(DEV)
    first = _rand((1, N, C), 64).to(DEV)
    dims = head.mask_head[0].make_dims(T, N, H, W)
    packs = [h.stage_pack(torch.device(DEV)) for h in head.mask_head]
    xh = xs.to(dt)
    a = vkn.ops.head_forward(dims, packs, xh, pfs, mps, None, case['up'], clip_first_prev=first)
    b = vkn.ops.head_forward(dims, packs, xh.float(), pfs, mps, None, case['up'], clip_first_prev=first)
    for u, v in zip(a, b):
        assert (u is None and v is None) or torch.equal(u, v)
    if (H * W) % 64 == 0:
        c = vkn.ops.head_forward(dims, packs, xh, pfs, mps, None, case['up'], clip_first_prev=first, flags=vkn.ops.FLAG_BITS_HANDOFF)
        for u, v in zip(a, c):
            assert (u is None and v is None) or torch.equal(u, v)
    one = [packs[0]]
    ra = vkn.ops.head_forward(dims, one, xh, pfs, mps, None, 1)
    rb = vkn.ops.head_forward(dims, one, xs, pfs, mps, None, 1)
    scale = rb[2].abs().max().item()
    assert (ra[2] - rb[2]).abs().max().item() <= tol * scale, ((ra[2] - rb[2]).abs().max().item(), scale)


def test_half_x_single_stage_and_class_api(vkn):
    """`KernelUpdateHead.forward` (vkn_stage_forward_f32) and the class-level fused head take half-storage x as is."""
    from test_gpu_parity import _build_head
    _, case = load_golden('video_cfg')
    head, _ = _build_head(vkn, case)
    T, N, C, H, W = 2, case['N'], case['C'], case['H'], case['W']
    xs = _clamp_tiny(_rand((T, C, H, W), 51)).to(DEV)
    pfs = _rand((T, N, C, 1, 1), 52).to(DEV)
    mps = _rand((T, N, H, W), 53, 4.0).to(DEV)
    xh = xs.to(torch.bfloat16)
    head.eval()
    with torch.no_grad():
        a = head.mask_head[0](xh, pfs, mps)
        b = head.mask_head[0](xh.float(), pfs, mps)
        for u, v in zip(a, b):
            assert (u is None and v is None) or torch.equal(u, v)
        a = head.simple_test_mask_preds(xh, pfs, mps, None, None)
        b = head.simple_test_mask_preds(xh.float(), pfs, mps, None, None)
        for u, v in zip(a, b):
            assert (u is None and v is None) or torch.equal(u, v)
    # the autograd (training) path takes half-storage features too since round 5 (it raised TypeError before): forward outputs are the
    # bits of the fp32 path on the rounded x
    a = head.mask_head[0](xh, pfs, mps)
    b = head.mask_head[0](xh.float(), pfs, mps)
    for u, v in zip(a, b):
        assert (u is None and v is None) or torch.equal(u, v)
    with pytest.raises(TypeError):
        head.mask_head[0](xh.double(), pfs, mps)


def test_half_x_rejected_by_reference_kernels_and_ragged_sizes(vkn):
    x = _rand((1, 64, 6, 10), 41).to(DEV).half()
    masks = _rand((1, 32, 6, 10), 42).to(DEV)
    with pytest.raises(vkn._lib.VknError):
        vkn.ops.mask_gather(x, masks)            # H*W % 64 != 0
    x = _rand((1, 64, 8, 8), 43).to(DEV).half()
    masks = _rand((1, 32, 8, 8), 44).to(DEV)
    with pytest.raises(vkn._lib.VknError):
        vkn.ops.mask_gather(x, masks, flags=vkn.ops.FLAG_REF_KERNELS)


@pytest.mark.parametrize('dt', [torch.float16, torch.bfloat16], ids=['fp16', 'bf16'])
@pytest.mark.parametrize('sem', [True, False], ids=['semantic_fpn', 'loc_only'])
def test_kernel_init_pass_on_half_storage_features(vkn, dt, sem):
    """The kernel-initialisation pass (`ConvKernelHead._decode_init_proposals`, knet/det/kernel_head.py:204-263) on fp16 / bf16 feature
    STORAGE (VERDICT r03 "missing" 4): mask_preds / seg_preds are the BITS of the fp32 pass on the widened features, x_feats is
    half(float(sem) + float(loc)) exactly, and proposal_feats = init kernels + the fp32 gather of that x_feats, bit for bit."""
    B, C, H, W, Np, ncls, nth = 2, 256, 32, 64, 100, 19, 8
    loc = _clamp_tiny(_rand((B, C, H, W), 71)).to(DEV).to(dt)
    semf = _clamp_tiny(_rand((B, C, H, W), 72)).to(DEV).to(dt) if sem else None
    init_w = _rand((Np, C, 1, 1), 73, 0.1).to(DEV)
    seg_w = _rand((ncls, C, 1, 1), 74, 0.1).to(DEV) if sem else None
    seg_b = _rand((ncls,), 75, 0.1).to(DEV) if sem else None
    kw = dict(num_thing_classes=nth, cat_stuff_mask=sem, proposal_feats_with_obj=True)
    prop, xf, masks, seg = vkn.ops.kernel_init(loc, semf, init_w, seg_w, seg_b, **kw)
    prop32, xf32, masks32, seg32 = vkn.ops.kernel_init(loc.float(), semf.float() if sem else None, init_w, seg_w, seg_b, **kw)
    assert xf.dtype == dt
    assert torch.equal(masks, masks32)
    if sem:
        assert torch.equal(seg, seg32)
        assert torch.equal(xf, (loc.float() + semf.float()).to(dt))
    else:
        assert torch.equal(xf, loc)
    xraw, _ = vkn.ops.mask_gather(xf.float(), masks[:, :Np].contiguous(), 0.5)
    assert torch.equal(prop[:, :Np], init_w.reshape(1, Np, C) + xraw)
    if sem:
        assert torch.equal(prop[:, Np:], seg_w.reshape(ncls, C)[nth:].unsqueeze(0).expand(B, -1, -1))
    # the head takes that x_feats as is: same bits as on the widened copy
    assert torch.equal(vkn.ops.mask_gather(xf, masks, 0.5)[0], vkn.ops.mask_gather(xf.float(), masks, 0.5)[0])


@pytest.mark.parametrize('S,H,W', [(4, 128, 256), (2, 48, 80), (4, 92, 160), (4, 9, 15)])
def test_upsample_with_fp16_output_is_the_fp32_result_rounded_once(vkn, S, H, W):
    """VKN_FLAG_SCALED_F16 / vkn_upsample_bilinear_f16out (opt-in, VERDICT r04 item 9: "offer fewer bytes"): the up-scaled logits as
    fp16.  Stated tolerance: |error| <= 2^-11 |logit| — because the kernel interpolates in fp32 exactly as the fp32 kernel does and
    rounds once at the store, its output is BIT-IDENTICAL to `upsample_bilinear(z).half()`; the binary mask (sign) is unchanged."""
    g = torch.Generator().manual_seed(5)
    z = (torch.randn(2, 7, H, W, generator=g) * 6.0).to(DEV)
    ref = vkn.ops.upsample_bilinear(z, S)
    if (W * S) % 4:
        with pytest.raises(vkn.VknError):
            vkn.ops.upsample_bilinear(z, S, out_f16=True)
        return
    got = vkn.ops.upsample_bilinear(z, S, out_f16=True)
    assert got.dtype == torch.float16 and got.shape == ref.shape
    assert torch.equal(got, ref.half())
    assert float((got.float() - ref).abs().max()) <= 2.0 ** -11 * float(ref.abs().max())
    assert torch.equal(got > 0, ref > 0)


def test_head_forward_with_fp16_scaled_output(vkn):
    """The fused head call with VKN_FLAG_SCALED_F16: every output but `scaled` is bit-identical to the plain call, `scaled` is the plain
    call's fp32 result rounded to fp16."""
    from test_gpu_parity import _build_head, _cuda
    g, case = load_golden('video_cfg')
    head, (x, pf, mp, prev) = _build_head(vkn, case)
    xd, pfd, mpd, prevd = _cuda(x, pf, mp, prev)
    with torch.no_grad():
        a = head._head_forward(xd, pfd, mpd, prevd, want_track=True)
        b = head._head_forward(xd, pfd, mpd, prevd, want_track=True, flags=vkn.ops.FLAG_SCALED_F16)
    for k in (0, 1, 2, 4):
        assert torch.equal(a[k], b[k]), k
    assert b[3].dtype == torch.float16 and torch.equal(b[3], a[3].half())


@pytest.mark.parametrize('dt', [torch.float16, torch.bfloat16], ids=['fp16', 'bf16'])
def test_forward_train_on_half_storage_features(vkn, dt, name='train_cfg'):
    """the TRAINING path on fp16 / bf16 feature storage (BASELINE cfg2 words "bf16", cfg5 "fp16").  Same pin as the
    inference path: on x' = float(half(x)) the fp32 path returns the SAME losses bit for bit (its forward kernels are the half-storage
    kernels with the all-zero low-half terms), the parameter / kernel gradients are the same bits (dK runs on the widened x in both
    cases); x.grad arrives in x's storage type (each of the six contributions rounded once from its fp32 value)."""
    from test_gpu_train import _train_case
    g, case, head, (x, pf, mp, prev), (gt_masks, gt_labels, gt_sem_seg, gt_sem_cls) = _train_case(vkn, name)
    assert (case['H'] * case['W']) % 64 == 0 and case['C'] in (64, 128, 256)
    metas = [dict() for _ in range(case['B'])]
    xh = _clamp_tiny(x).to(dt)                      # (bf16 -> f16 inside the kernels is exact in the normal f16 range)

    def run(xin):
        xd = xin.to(DEV).requires_grad_(True)
        pfd = pf.to(DEV).requires_grad_(True)
        for p in head.parameters():
            p.grad = None
        if case['video']:
            out = head.forward_train_with_previous(xd, pfd, mp.to(DEV), None, metas, gt_masks, gt_labels, gt_sem_seg=gt_sem_seg,
                                                   gt_sem_cls=gt_sem_cls, previous_obj_feats=prev.to(DEV))
            losses, extra = out[0], 0.01 * (out[5] ** 2).sum()
        else:
            losses, extra = head.forward_train(xd, pfd, mp.to(DEV), None, metas, gt_masks, gt_labels, gt_sem_seg=gt_sem_seg, gt_sem_cls=gt_sem_cls), 0.0
        total = sum(v for k, v in losses.items() if 'loss' in k) + extra
        total.backward()
        return ({k: float(v) for k, v in losses.items()}, xd.grad.clone(), pfd.grad.clone(),
                {n: p.grad.clone() for n, p in head.named_parameters() if p.grad is not None})

    l32, gx32, gpf32, gp32 = run(xh.float())
    l16, gx16, gpf16, gp16 = run(xh)
    assert l16 == l32, (l16, l32)
    # x.grad: x feeds six differentiable ops (a gather and a decode per stage); each hands back its fp32 gradient rounded ONCE to x's
    # type and autograd accumulates the six in that type — so not the fp32 sum rounded once, but within a few units of x's precision
    eps = 2.0 ** -11 if dt == torch.float16 else 2.0 ** -8
    assert gx16.dtype == dt and float((gx16.float() - gx32).abs().max()) <= 6 * eps * float(gx32.abs().max())
    assert torch.equal(gpf16, gpf32)
    assert gp16.keys() == gp32.keys() and all(torch.equal(gp16[k], gp32[k]) for k in gp16)


def test_forward_train_on_fp16_storage_at_the_benchmarked_cfg3_size(vkn):
    """the cfg3-size training step (128x256 features, x4, C = 256, N = 117, video head — the golden
    `train_video_cfg3`) with x STORED as fp16: same losses bit for bit and same parameter / kernel gradients bit for bit as the fp32 path on
    the rounded x, x.grad in fp16 — the pin that carries the reference golden of the fp32 path over to half storage at this size."""
    test_forward_train_on_half_storage_features(vkn, torch.float16, name='train_video_cfg3')


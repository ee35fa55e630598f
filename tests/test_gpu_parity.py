"""GPU (-m gpu): parity of the HIP path (through the C ABI) against the CPU oracle on the same seeded inputs, against the
golden vectors captured from the reference, and — at BASELINE cfg2 size — through size-independent properties.

Tolerances (north_star): mask logits within 1e-3 (fp32); integer artefacts (binary masks, pixel counts) bit-exact.
"""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from helpers import record_margins, GOLDEN, cfg_of, load_golden, make_case, maxabs, run_oracle
from oracle import knet_oracle as O
from oracle import synth

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
TOL_LOGIT = 1e-3


def _cuda(*ts):
    return [t.to(DEV) if t is not None else None for t in ts]


def _rand(shape, salt, std=1.0):
    return torch.from_numpy(synth.normalish(shape, salt, std))


SHAPES = [  # B, N, C, H, W
    (2, 15, 64, 8, 16),      # tiny
    (1, 21, 64, 9, 15),      # ragged P = 135 (scalar-load path, pixel tail)
    (2, 117, 256, 16, 32),   # config channels / kernels
    (1, 117, 256, 64, 128),  # BASELINE cfg1 size
    (1, 166, 256, 23, 40),   # VIP-Seg kernel count: two n-chunks (128 + 64 rows), P % 32 != 0
    (3, 100, 32, 4, 8),      # one channel block, one pixel tile
    (2, 40, 64, 16, 16),     # P % 128 == 0, C % 64 == 0: the 16-byte decode kernel (k_decode4), two n-blocks
    (1, 166, 128, 8, 32),    # k_decode4 / fused pass with two n-chunks (128 + 64 rows), C = 128
]


def test_library_loaded_and_gpu_visible(vkn):
    assert torch.cuda.is_available()
    assert os.path.exists(vkn._lib.LIBPATH)
    assert vkn._lib.lib().vkn_version() == 0x000600


@pytest.mark.parametrize('flags', [0, 1], ids=['mfma', 'refkernels'])
@pytest.mark.parametrize('shape', SHAPES, ids=lambda s: 'x'.join(map(str, s)))
def test_mask_gather_vs_oracle(vkn, shape, flags):
    B, N, C, H, W = shape
    x, m = _rand((B, C, H, W), 101), _rand((B, N, H, W), 102, 4.0)
    xraw, cnt = vkn.ops.mask_gather(x.to(DEV), m.to(DEV), 0.5, flags)
    bits = O.binarize(m, 0.5)
    ref = O.mask_gather(x.double(), bits.double())
    assert torch.equal(cnt.cpu(), bits.sum(dim=(-1, -2))), 'ON-pixel counts must be bit-exact'
    scale = float(ref.abs().max()) + 1.0
    assert maxabs(xraw, ref) < 3e-6 * scale * max(1.0, np.sqrt(H * W) / 8)
    # and against the fp32 oracle op itself
    assert maxabs(xraw, O.mask_gather(x, bits)) < 2e-5 * scale


def test_threshold_bits_are_bit_exact(vkn):
    """(sigmoid(z) > 0.5) of the reference (torch CPU fp32) for every z of the KAT sweep, incl. 0, +-5e-8, 8.9e-8, +-1e-7."""
    g = np.load(os.path.join(GOLDEN, 'thr_kat.npz'))
    z, bit = g['z'], g['bit']
    n = len(z)
    rows = 256
    B = (n + rows - 1) // rows
    zz = np.full(B * rows, -1.0, dtype=np.float32)
    zz[:n] = z
    m = torch.from_numpy(zz).reshape(B, rows, 1, 1).expand(B, rows, 4, 8).contiguous()
    x = torch.ones(B, 32, 4, 8)
    _, cnt = vkn.ops.mask_gather(x.to(DEV), m.to(DEV), 0.5)
    got = (cnt.cpu().reshape(-1)[:n] == 32).numpy()
    assert set(np.unique(cnt.cpu().numpy())) <= {0.0, 32.0}
    assert np.array_equal(got, bit)


@pytest.mark.parametrize('flags', [0, 1], ids=['mfma', 'refkernels'])
@pytest.mark.parametrize('shape', SHAPES, ids=lambda s: 'x'.join(map(str, s)))
def test_mask_decode_vs_oracle(vkn, shape, flags):
    B, N, C, H, W = shape
    x, k, bias = _rand((B, C, H, W), 201), _rand((B, N, C), 202, 0.7), _rand((B, N), 203)
    out = vkn.ops.mask_decode(x.to(DEV), k.to(DEV), bias.to(DEV), flags)
    ref = O.mask_decode(x.double(), k.double().reshape(B, N, C, 1, 1), 1) + bias.double()[:, :, None, None]
    assert maxabs(out, ref) < 1e-4, 'decode logits (|logit| ~ 30) must stay far inside the 1e-3 budget'
    out0 = vkn.ops.mask_decode(x.to(DEV), k.to(DEV), None, flags)
    assert maxabs(out0, ref - bias.double()[:, :, None, None]) < 1e-4


def test_decode_planes_entry_matches(vkn):
    B, N, C, H, W = 2, 117, 256, 16, 32
    x, k = _rand((B, C, H, W), 211).to(DEV), _rand((B, N, C), 212, 0.7).to(DEV)
    hi, lo = vkn.ops.split_planes(k)
    assert maxabs(hi.float() + lo.float(), torch.cat([k, torch.zeros(B, 128 - N, C, device=DEV)], 1)) < 2e-6
    assert torch.equal(vkn.ops.mask_decode_planes(x, hi, lo, N), vkn.ops.mask_decode(x, k))


@pytest.mark.parametrize('shape', [(2, 40, 64, 16, 16), (1, 166, 128, 8, 32), (2, 117, 256, 16, 32), (3, 117, 256, 24, 40)],
                         ids=lambda s: 'x'.join(map(str, s)))
def test_fused_decode_gather_is_decode_then_gather(vkn, shape):
    """k_fused_dg (stage s decode + stage s+1 gather in one pass over x) == k_decode* followed by k_gather*, BIT for bit:
    same per-element MFMA sequence for the logits, same tile order / partial sums / fixed-order reduce for the gather."""
    B, N, C, H, W = shape
    x, k, bias = _rand((B, C, H, W), 221).to(DEV), _rand((B, N, C), 222, 0.25).to(DEV), _rand((B, N), 223).to(DEV)
    hi, lo = vkn.ops.split_planes(k)
    logits = vkn.ops.mask_decode_planes(x, hi, lo, N, bias)
    xr_ref, cnt_ref = vkn.ops.mask_gather(x, logits)
    xr, cnt = vkn.ops.decode_gather(x, hi, lo, N, bias)
    assert torch.equal(cnt, cnt_ref) and torch.equal(xr, xr_ref)
    assert 0.05 < float((cnt > 0).float().mean())  # not a vacuous case
    # and against fp64 on the device
    M = (logits >= vkn.ops.thr_logit(0.5)).double()
    ref = torch.bmm(M.reshape(B, N, -1), x.double().reshape(B, C, -1).transpose(1, 2))
    assert maxabs(xr, ref) < 2e-5 * max(1.0, float(ref.abs().max()))


@pytest.mark.parametrize('scale', [2, 4])
def test_upsample_vs_torch(vkn, scale):
    m = _rand((2, 7, 9, 15), 301, 5.0)
    out = vkn.ops.upsample_bilinear(m.to(DEV), scale)
    ref = F.interpolate(m, scale_factor=scale, mode='bilinear', align_corners=False)
    assert out.shape == ref.shape and maxabs(out, ref) < 2e-5


def test_kernel_updator_vs_oracle(vkn):
    C, M = 64, 30
    ku = vkn.build_transformer_layer(dict(type='KernelUpdator', in_channels=C, feat_channels=C, out_channels=C))
    shapes = {k: tuple(v.shape) for k, v in ku.state_dict().items()}
    sd = {k: torch.from_numpy(v) for k, v in synth.state_dict_like(shapes, 7).items()}
    ku.load_state_dict(sd)
    ku = ku.to(DEV).eval()
    u, k = _rand((2, 15, C), 401, 20.0), _rand((2, 15, 1, C), 402)
    with torch.no_grad():
        out = ku(u.to(DEV), k.to(DEV))
    ref = O.kernel_updator({'ku.' + n: v for n, v in sd.items()}, 'ku', u, k, O.HeadCfg(in_channels=C, feat_channels=C))
    assert out.shape == ref.shape and maxabs(out, ref) < 2e-5


def _build_head(vkn, case):
    from test_host_logic import _cfg
    over = dict(previous_link=case['plink'], previous_type=case['ptype']) if 'plink' in case else None
    head = vkn.build_head(_cfg(bool(case['video']), C=case['C'], heads=case['heads'], ffn=case['ffn'], ncls=case['ncls'],
                               n_thing=case['n_thing'], n_stuff=case['n_stuff'], S=case['S'], up=case['up'],
                               nprop=case['nprop'], mask_over=over))
    cfg, sd, x, pf, mp, prev = make_case(case)
    head.load_state_dict(sd, strict=True)
    return head.to(DEV).eval(), (x, pf, mp, prev)


@pytest.mark.parametrize('name', ['det_tiny', 'det_odd', 'det_cfg', 'video_tiny', 'video_cfg'])
def test_stage_by_stage_vs_oracle_and_reference(vkn, name):
    """`_mask_forward` per stage (the reference's own per-stage API) against the oracle trace AND the reference goldens."""
    g, case = load_golden(name)
    head, (x, pf, mp, prev) = _build_head(vkn, case)
    traces = []
    run_oracle(case, traces=traces)
    xd, obj, masks, prevd = _cuda(x, pf, mp, prev)
    with torch.no_grad():
        for s in range(case['S']):
            kw = {}
            if case['video'] and s == case['S'] - 1:
                kw = dict(previous_obj_feats=prevd)
            r = head._mask_forward(s, xd, obj, masks, [dict()] * case['B'], **kw)
            obj, masks = r['object_feats'], r['mask_preds']
            tr = traces[s]
            assert maxabs(r['cls_score'], tr['cls_score']) < 1e-4, f'stage {s} cls'
            assert maxabs(obj, tr['obj_feat']) < 1e-4, f'stage {s} obj'
            assert maxabs(masks, tr['new_mask_preds']) < TOL_LOGIT, f'stage {s} masks'
            assert maxabs(r['cls_score'], g[f's{s}_cls']) < 1e-4 and maxabs(obj, g[f's{s}_obj']) < 1e-4
            if 'x_feats' in r:
                assert maxabs(r['x_feats'], tr['x_feat']) < 1e-3 * (1 + float(tr['x_feat'].abs().max()))
        if case['video']:
            assert maxabs(r['object_feats_track'], g['track']) < 1e-4


def _chain_flags(vkn, chain):
    """the forms of the [N x C] chain by name; `persistent`: the persistent kernels as shipped (two-term fp16 split, vkn_chain_h2.hip); `persistent_bf16x3`: on the three-term bf16 split"""
    o = vkn.ops
    return dict(persistent=o.FLAG_CHAIN_PERSISTENT, launches=o.FLAG_CHAIN_LAUNCHES, ksplit=o.FLAG_CHAIN_KSPLIT,
                persistent_bf16x3=o.FLAG_CHAIN_PERSISTENT | o.FLAG_CHAIN_BF16X3)[chain]


@pytest.mark.parametrize('flags', [0, 1, 2, 3, 512, 513, 256, 8192, 512 + 65536], ids=['mfma', 'refkernels', 'exactgemm', 'allexact', 'persistent', 'persistent_ref', 'launches', 'ksplit', 'persistent_bf16x3'])
@pytest.mark.parametrize('name', ['det_tiny', 'det_odd', 'det_cfg', 'video_tiny', 'video_cfg'])
def test_head_vs_reference_golden(vkn, name, flags):
    """The fused S-stage call (`simple_test_mask_preds[_plus_previous]`) against the REFERENCE's own outputs."""
    g, case = load_golden(name)
    head, (x, pf, mp, prev) = _build_head(vkn, case)
    xd, pfd, mpd, prevd = _cuda(x, pf, mp, prev)
    with torch.no_grad():
        obj, cls, masks, scaled, track = head._head_forward(xd, pfd, mpd, prevd if case['video'] else None,
                                                            want_track=bool(case['video']), flags=flags)
    assert maxabs(obj, g['object_feats']) < 1e-4
    assert maxabs(cls, g['cls_score']) < 1e-5
    assert maxabs(masks, g['mask_preds']) < TOL_LOGIT
    ref_bits = g['mask_preds'] > 0
    margin = np.abs(g['mask_preds']) > 2e-3
    assert np.array_equal((masks.cpu().numpy() > 0)[margin], ref_bits[margin]), 'binary masks must be bit-exact'
    up = case['up']
    assert tuple(scaled.shape[-2:]) == (case['H'] * up, case['W'] * up)
    if 'scaled_mask_preds' in g:
        assert maxabs(scaled, g['scaled_mask_preds']) < TOL_LOGIT
    else:
        # row sums of the upsampled logits, tolerance relative to the row's ABSOLUTE mass (sums themselves cancel to ~0)
        rs = scaled.double().sum(dim=(-1, -2)).cpu().numpy()
        mass = np.abs(g['mask_preds']).sum(axis=(-1, -2)) * up * up
        assert np.max(np.abs(rs - g['scaled_rowsum']) / mass) < 1e-5
    if case['video']:
        assert maxabs(track, g['track']) < 1e-4
    # public API returns
    with torch.no_grad():
        if case['video']:
            out = head.simple_test_mask_preds_plus_previous(xd, pfd, mpd, None, [dict()] * case['B'], previous_obj_feats=prevd)
        else:
            out = head.simple_test_mask_preds(xd, pfd, mpd, None, [dict()] * case['B'])
    assert len(out) == 4 and tuple(out[0].shape) == (case['B'], case['N'], case['C'], 1, 1)
    if flags == 0:
        assert torch.equal(out[2], masks), 'same inputs -> bit-identical outputs (deterministic kernels)'


def test_head_cfg1_size_vs_reference_golden(vkn):
    g, case = load_golden('det_cfg_big')
    head, (x, pf, mp, _) = _build_head(vkn, case)
    with torch.no_grad():
        obj, cls, masks, scaled = head.simple_test_mask_preds(*_cuda(x, pf, mp), None, [dict()])
    assert maxabs(obj, g['object_feats']) < 1e-4 and maxabs(cls, g['cls_score']) < 1e-5
    flat = masks.reshape(-1).cpu()
    assert maxabs(flat[torch.from_numpy(g['sample_idx'])], g['sample_val']) < TOL_LOGIT
    rs = masks.double().sum(dim=(-1, -2)).cpu().numpy()
    assert np.max(np.abs(rs - g['mask_rowsum'])) < 1e-4 * np.max(g['mask_rowabs'])
    bits = np.packbits(flat.numpy() > 0)
    assert np.all((bits ^ g['sign_bits']) & g['sign_valid'] == 0), 'binary masks (|logit| > 2e-3) must be bit-exact'


# Measured on MI355X in round 4 (profiles/r04_parity_margins.json) over FOUR realisations of the [N x C] chain's fp32 summation order
# (launch-per-GEMM on k_gemm_s3 / on k_gemm_t3, the persistent kernels, the persistent kernels with a split FFN), then given 2x
# head-room: errors may double (capped by the hard tolerance), a share may lose as many rows again as it has lost (at least two),
# flipped bits may double (at least two).  Each of the two VIP-Seg cases has ONE kernel whose hand-over mask sits on the threshold: it
# flips in some realisations and not in others (worst seen: 1 row, 7 off-threshold bits, clean kernels 4.9e-5, clean logits 6.6e-4;
# best: every row clean, 0 bits, 7.0e-6 / 5.9e-5).  YouTube-VIS: every row clean in every realisation.
# Round 5: the limits are anchored on the REFERENCE ALGORITHM'S OWN re-ordering noise (tools/oracle_reorder_noise.py ->
# profiles/r05_oracle_reorder_noise.json: the CPU oracle against the reference goldens with nothing changed but the fp32 summation
# order — permuted input channels of the 1x1 conv, 1 / 16 intra-op threads).  video_vipseg_big: the oracle itself lands at 0.964 of the
# rows within 2e-4 and 39 wrong off-threshold bits with ONE thread (1.000 / 0 with eight); video_vipseg_n216 (92x160 since round 5):
# 0.9815 .. 1.0 / 0 .. 41 bits over six re-orderings — our three chain forms measure 0.9815 / 41, the SAME pixels.  Round 4
# had set "2 x what our two chain forms happened to measure" (14 bits), which the third form (few-row chain: 0.958 / 45 bits, worst
# clean-row errors 5.8e-5 / 5.7e-4 — and the smallest error of all forms against fp64, tools/chain_accuracy.py) missed by chance, exactly
# like the reference with another thread count would.  Limits = the worse of (our forms, the oracle's re-orderings) with ~2x head-room
# on counts; the clean-row error limits stay at the parity tolerances.
FREE_RUN_LIMITS = {
    'video_vipseg_big': dict(clean_share_of_stable=0.985, share_rows_kernels_within_2e4=0.93, share_rows_sampled_logits_within_1e3=0.93,
                             wrong_bits_off_threshold=90, worst_clean_kernel_err=1.2e-4, worst_clean_sampled_logit_err=1.0e-3),
    'det_ytvis': dict(clean_share_of_stable=0.99, share_rows_kernels_within_2e4=1 - 2 / 200, share_rows_sampled_logits_within_1e3=1 - 2 / 200,
                      wrong_bits_off_threshold=2, worst_clean_kernel_err=1.2e-5, worst_clean_sampled_logit_err=1.0e-4),
    'video_vipseg_n216': dict(clean_share_of_stable=0.99, share_rows_kernels_within_2e4=0.96, share_rows_sampled_logits_within_1e3=0.96,
                              wrong_bits_off_threshold=90, worst_clean_kernel_err=1.2e-4, worst_clean_sampled_logit_err=1.0e-3),
}


@pytest.mark.parametrize('chain', ['auto', 'launches', 'persistent', 'persistent_bf16x3'])
@pytest.mark.parametrize('name', ['video_vipseg_big', 'det_ytvis', 'video_vipseg_n216'])
def test_head_cfg5_cfg4_size_vs_reference_golden(vkn, name, chain):
    """(`video_vipseg_n216`: BASELINE cfg5 as LITERALLY worded — 150 proposals + 66 stuff kernels = 216 rows, 92x160 features (round 5; 46x80 before).)
    BASELINE cfg5 at its real size (video_knet_s3_swinb VIP-Seg: N = 166 = 100 + 66 kernels -> two n-chunks, C = 256, 92x160
    features, 124 classes, x4, tracking link) and the cfg4 per-frame shape (YouTube-VIS: N = 100, 48x80, 40 thing classes, no
    stuff, x2, 2 frames): the free-running 3-stage fused head against the REFERENCE's own outputs."""
    g, case = load_golden(name)
    head, (x, pf, mp, prev) = _build_head(vkn, case)
    if chain != 'auto':          # (default at these sizes: the few-row chain, vkn_ksplit.hip; all forms of the chain are held to the same bounds)
        for h in head.mask_head:
            h.vkn_flags = _chain_flags(vkn, chain)
    metas = [dict()] * case['B']
    B, N, P = case['B'], case['N'], case['H'] * case['W']
    # kernels whose hand-over masks stay clear of the binarisation threshold in EVERY stage of the reference cannot flip a bit under
    # a different fp32 summation order (golden `row_margin`); the few others may (chaos note) and get a loose bound
    stable = torch.from_numpy((g['row_margin'] > 5e-5).all(axis=0))          # [B, N]
    assert float(stable.float().mean()) > 0.8

    with torch.no_grad():
        if case['video']:
            obj, cls, masks, scaled, track = head.simple_test_mask_preds_plus_previous(
                *_cuda(x, pf, mp), None, metas, previous_obj_feats=prev.to(DEV), return_track=True)
        else:
            obj, cls, masks, scaled = head.simple_test_mask_preds(*_cuda(x, pf, mp), None, metas)
            track = None
    # `row_margin > 5e-5` keeps the REFERENCE's logits of a stable kernel that far from the threshold — but our logits may differ from
    # the reference's by as much (5e-5 is the typical deviation of the kernels here), so a "stable" kernel can still flip a pixel
    # when a summation order changes (one did when the attention moved to the matrix cores: its features move by ~5e-3 and, through
    # the attention, every kernel of its frame inherits a share).  Hence a flip budget, as in the cfg2 test below: the CLEAN rows
    # are the stable rows whose final kernels are within 2e-4 of the reference — at least 98 % of the stable rows — and they meet
    # every tight bound; all rows meet the loose ones.  (Each stage on its own is bounded by the teacher-forced tests.)
    d_obj = (obj.cpu() - torch.from_numpy(g['object_feats'])).abs().reshape(B, N, -1).amax(-1)
    clean = stable & (d_obj < 2e-4)
    assert float(clean.sum()) >= 0.98 * float(stable.sum()), (int(clean.sum()), int(stable.sum()))
    assert float(d_obj.max()) < 5e-2
    if track is not None:
        d = (track.cpu() - torch.from_numpy(g['track'])).abs().reshape(B, N, -1).amax(-1)
        assert float(d[clean].max()) < 1e-2 and float(d.max()) < 5e-2      # (the link attends over ALL kernels of two frames: coupled)
        assert float((d[clean] < 2e-4).float().mean()) > 0.9
    d = (cls.cpu() - torch.from_numpy(g['cls_score'])).abs().amax(-1)
    assert float(d[clean].max()) < 1e-4 and float(d.max()) < 1e-2
    flat = masks.reshape(-1).cpu()
    idx = torch.from_numpy(g['sample_idx'])
    srow = clean.reshape(-1)[idx // P]                                         # the (frame, kernel) row of each sampled logit
    d = (flat[idx] - torch.from_numpy(g['sample_val'])).abs()
    assert float(d[srow].max()) < TOL_LOGIT and float(d.max()) < 0.5
    rs = masks.double().sum(dim=(-1, -2)).cpu()
    drs = (rs - torch.from_numpy(g['mask_rowsum'])).abs()
    assert float(drs[clean].max()) < 1e-4 * np.max(g['mask_rowabs'])
    bits = np.unpackbits(np.packbits(flat.numpy() > 0) ^ g['sign_bits'])[:flat.numel()] & np.unpackbits(g['sign_valid'])[:flat.numel()]
    wrong = torch.from_numpy(bits.astype(bool)).reshape(B, N, P)
    assert not bool(wrong[clean].any()), 'binary masks (|logit| > 2e-3) of the clean kernels must be bit-exact'
    assert float(wrong.float().mean()) < 1e-4
    # ---- what was measured: shares and worst errors, recorded beside the pass / fail
    per_row = torch.zeros(B * N).scatter_reduce(0, idx // P, (flat[idx] - torch.from_numpy(g['sample_val'])).abs(), 'amax', include_self=True)
    sampled = torch.zeros(B * N, dtype=torch.bool).index_fill_(0, idx // P, True)
    dcls = (cls.cpu() - torch.from_numpy(g['cls_score'])).abs().amax(-1)
    m = dict(rows=B * N, stable_share=float(stable.float().mean()), clean_share_of_stable=float(clean.sum()) / float(stable.sum()),
             share_rows_kernels_within_2e4=float((d_obj < 2e-4).float().mean()),
             share_rows_sampled_logits_within_1e3=float((per_row[sampled] < 1e-3).float().mean()),
             worst_clean_kernel_err=float(d_obj[clean].max()), worst_any_kernel_err=float(d_obj.max()),
             worst_clean_cls_err=float(dcls[clean].max()), worst_any_cls_err=float(dcls.max()),
             worst_clean_sampled_logit_err=float(d[srow].max()), worst_any_sampled_logit_err=float(d.max()),
             worst_clean_rowsum_rel=float(drs[clean].max() / np.max(g['mask_rowabs'])),
             wrong_bits_off_threshold=int(wrong.sum()), wrong_bit_share=float(wrong.float().mean()))
    if track is not None:
        dt = (track.cpu() - torch.from_numpy(g['track'])).abs().reshape(B, N, -1).amax(-1)
        m.update(worst_clean_track_err=float(dt[clean].max()), share_clean_track_within_2e4=float((dt[clean] < 2e-4).float().mean()))
    record_margins(f'head_free_running_vs_reference[{name}-{chain}]', m)
    # tightened to the measured values with 2x head-room (profiles/r04_parity_margins.json); the loose bounds above remain as the
    # hard limits a flipped near-threshold pixel may reach
    lim = FREE_RUN_LIMITS[name]
    assert m['clean_share_of_stable'] >= lim['clean_share_of_stable'] and m['share_rows_kernels_within_2e4'] >= lim['share_rows_kernels_within_2e4']
    assert m['share_rows_sampled_logits_within_1e3'] >= lim['share_rows_sampled_logits_within_1e3']
    assert m['wrong_bits_off_threshold'] <= lim['wrong_bits_off_threshold']
    assert m['worst_clean_kernel_err'] <= lim['worst_clean_kernel_err'] and m['worst_clean_sampled_logit_err'] <= lim['worst_clean_sampled_logit_err']


# measured, round 4 (profiles/r04_parity_margins.json): 9 / 13 flipped bits over the three stages and 108 / 104 of 117 rows clean at the
# end (launch-per-GEMM / persistent chain), worst clean-row logit error 3.7e-4 / 4.9e-4.  Round 5 measured the REFERENCE ALGORITHM against
# itself on this very case (tools/oracle_reorder_noise.py, profiles/r05_oracle_reorder_noise.json): three channel permutations of the 1x1
# conv and 1 / 16 threads give 5 .. 24 flipped bits, 93 .. 112 clean rows, worst clean-row logit error 3.2e-4 .. 8.3e-4 — our forms sit
# INSIDE the band a re-ordered reference spans.  Limits: 2 x the oracle's worst flip count, its worst clean-row count minus head-room.
CFG2_FLIP_LIMIT, CFG2_CLEAN_ROWS_MIN, CFG2_CLEAN_LOGIT_ERR = 48, 88, 1.0e-3


@pytest.mark.parametrize('chain', ['auto', 'launches', 'persistent', 'persistent_bf16x3'])
def test_cfg2_size_free_running_vs_oracle_flip_budget(vkn, chain):
    """The FREE-RUNNING 3-stage head at BASELINE cfg2 size against the free-running oracle, with the chaos argument measured
    instead of assumed: per stage, the binarised masks may differ from the oracle's only where the oracle's logit
    is within 5e-4 of the threshold (half the 1e-3 logit budget), and every kernel row whose mask has no flipped bit stays within
    1e-3 (logits) / 2e-4 (kernels) of the oracle.  The measured quantities (flipped bits, clean rows, worst clean-row errors per
    stage) are recorded (`record_margins`) and bounded by twice their round-4 values: 26 of the 3 x 3.8 M bits."""
    case = dict(C=256, heads=8, ffn=2048, ncls=19, n_thing=2, n_stuff=17, S=3, up=4, nprop=100, N=117, H=128, W=256,
                B=1, seed=12, video=0)
    head, (x, pf, mp, _) = _build_head(vkn, case)
    if chain != 'auto':        # (auto at one frame: the few-row chain, vkn_ksplit.hip)
        for h in head.mask_head:
            h.vkn_flags = _chain_flags(vkn, chain)
    traces = []
    run_oracle(case, traces=traces)
    thr = vkn.ops.thr_logit(0.5)
    xd, o, m = x.to(DEV), pf.to(DEV), mp.to(DEV)
    clean = torch.ones(117, dtype=torch.bool)            # rows whose gather input never differed from the oracle's
    total_flips = 0
    rec = {}
    with torch.no_grad():
        for s in range(3):
            r = head._mask_forward(s, xd, o, m, [dict()])
            tr = traces[s]
            o, m = r['object_feats'], r['mask_preds']
            got, ref = m[0].cpu(), tr['new_mask_preds'][0]
            assert maxabs(r['object_feats'][0].reshape(117, -1).cpu()[clean], tr['obj_feat'][0].reshape(117, -1)[clean]) < 2e-4, s
            assert maxabs(got[clean], ref[clean]) < TOL_LOGIT, f'stage {s}: logits of kernels with unflipped masks'
            flip = (got >= thr) != (ref >= thr)
            fc = flip[clean]                              # kernels that entered this stage with oracle-identical masks
            nflip = int(fc.sum())
            total_flips += nflip
            assert nflip <= 64, f'stage {s}: {nflip} flipped bits in clean kernels'
            if nflip:
                assert float((ref[clean][fc] - thr).abs().max()) < 5e-4, 'only near-threshold logits (within half the 1e-3 budget) may flip'
            dl = (got - ref).abs().flatten(1).amax(1)
            do = (r['object_feats'][0].reshape(117, -1).cpu() - tr['obj_feat'][0].reshape(117, -1)).abs().amax(1)
            rec.update({f's{s}_clean_rows_in': int(clean.sum()), f's{s}_flipped_bits_clean': nflip, f's{s}_flipped_bits_all': int(flip.sum()),
                        f's{s}_flip_max_dist_to_thr': float((ref[clean][fc] - thr).abs().max()) if nflip else 0.0,
                        f's{s}_share_rows_logits_within_1e3': float((dl < 1e-3).float().mean()),
                        f's{s}_worst_clean_logit_err': float(dl[clean].max()), f's{s}_worst_clean_kernel_err': float(do[clean].max()),
                        f's{s}_worst_any_logit_err': float(dl.max())})
            clean = clean & ~flip.flatten(1).any(dim=1)   # the next stage gathers with these masks
    rec.update(clean_rows_out=int(clean.sum()), total_flipped_bits_clean=total_flips)
    record_margins(f'cfg2_free_running_vs_oracle[{chain}]', rec)
    assert int(clean.sum()) >= CFG2_CLEAN_ROWS_MIN, f'{int(clean.sum())} clean rows, {total_flips} flips'   # (the re-ordered oracle itself: 93 .. 112)
    # anchored on the reference's own re-ordering noise (profiles/r05_oracle_reorder_noise.json)
    assert total_flips <= CFG2_FLIP_LIMIT and int(clean.sum()) >= CFG2_CLEAN_ROWS_MIN
    assert all(rec[f's{s}_worst_clean_logit_err'] <= CFG2_CLEAN_LOGIT_ERR for s in range(3))


def test_clip_forward_matches_frame_by_frame(vkn):
    g, case = load_golden('video_tiny')
    head, (x, pf, mp, prev) = _build_head(vkn, case)
    xd, pfd, mpd, prevd = _cuda(x, pf, mp, prev)
    with torch.no_grad():
        obj, cls, masks, scaled, track = head.clip_forward(xd, pfd, mpd, first_previous_obj_feats=prevd[:1])
        # frame-by-frame with explicit previous = previous frame's final kernels
        chain = torch.cat([prevd[:1], obj[:-1]], 0)
        o2, c2, m2, s2, t2 = head._head_forward(xd, pfd, mpd, chain, want_track=True)
    assert torch.equal(masks, m2) and torch.equal(obj, o2) and maxabs(track, t2) < 1e-6
    # and the oracle agrees on the tracking embedding of frame 1 given frame 0's kernels
    cfg, sd, *_ = make_case(case)
    _, _, _, _, tr = O.iter_head_mask_preds(sd, x, pf, mp, cfg, previous_obj_feats=chain.cpu())
    assert maxabs(track, tr) < 1e-4


# ------------------------------------------------------------------------------------------------ BASELINE cfg2 size
CFG2 = dict(B=2, N=117, C=256, H=128, W=256)


def test_cfg2_size_properties(vkn):
    """1024x2048 frame -> 128x256 features: properties that need no CPU restatement at full size."""
    B, N, C, H, W = (CFG2[k] for k in 'BNCHW')
    g = torch.Generator().manual_seed(5)
    x = torch.randn(B, C, H, W, generator=g).to(DEV)
    k1, k2 = torch.randn(B, N, C, generator=g).to(DEV), torch.randn(B, N, C, generator=g).to(DEV)
    d1, d2, d12 = vkn.ops.mask_decode(x, k1), vkn.ops.mask_decode(x, k2), vkn.ops.mask_decode(x, k1 + k2)
    assert maxabs(d1 + d2, d12) < 2e-4, 'decode is linear in the kernels'
    assert torch.equal(d1, vkn.ops.mask_decode(x, k1)), 'deterministic'
    # selecting kernels reproduce feature channels: K[n] = e_{c(n)}
    sel = torch.zeros(B, N, C, device=DEV)
    idx = (torch.arange(N) * 2) % C
    sel[:, torch.arange(N), idx] = 1.0
    assert maxabs(vkn.ops.mask_decode(x, sel), x[:, idx]) < 2e-6
    # gather: all-ON mask = per-channel pixel sum and P counts; all-OFF = zeros; one-hot pixel masks pick x[:, :, p]
    on = torch.full((B, N, H, W), 3.0, device=DEV)
    xr, cnt = vkn.ops.mask_gather(x, on)
    ref = x.double().sum(dim=(-1, -2)).cpu()
    assert torch.all(cnt == H * W) and maxabs(xr[:, 0], ref) < 1e-3 and maxabs(xr[:, -1], ref) < 1e-3
    xr, cnt = vkn.ops.mask_gather(x, -on)
    assert float(xr.abs().max()) == 0.0 and float(cnt.abs().max()) == 0.0
    hot = torch.full((B, N, H * W), -5.0, device=DEV)
    pix = (torch.arange(N) * 277 + 13) % (H * W)
    hot[:, torch.arange(N), pix] = 5.0
    xr, cnt = vkn.ops.mask_gather(x, hot.reshape(B, N, H, W))
    assert torch.all(cnt == 1) and maxabs(xr, x.reshape(B, C, -1)[:, :, pix].transpose(1, 2)) < 2e-6
    # gather is the adjoint of decode:  <decode(x,K), M> == <K, gather(x,M)>  for binary M.  Tolerance: fp32-level
    # relative error on the ABSOLUTE mass of the two sums (the sums themselves are random-sign and cancel).
    mlog = torch.randn(B, N, H, W, generator=g).to(DEV)
    xr, cnt = vkn.ops.mask_gather(x, mlog)
    M = (mlog >= vkn.ops.thr_logit(0.5)).double()
    assert torch.equal(cnt.double(), M.sum(dim=(-1, -2)))
    lhs, rhs = (d1.double() * M).sum(), (k1.double() * xr.double()).sum()
    mass = float((k1.double().abs() * xr.double().abs()).sum())
    assert abs(float(lhs - rhs)) < 2e-6 * mass
    # full-size element-wise check against fp64 on the device (property check; the CPU oracle covers the small sizes)
    ref = torch.bmm(M.reshape(B, N, -1), x.double().reshape(B, C, -1).transpose(1, 2))
    assert maxabs(xr, ref) < 2e-5 * float(ref.abs().max())
    refd = torch.bmm(k1.double(), x.double().reshape(B, C, -1)).reshape(B, N, H, W)
    assert maxabs(d1, refd) < 2e-4


@pytest.mark.parametrize('chain', ['ksplit', 'launches', 'persistent', 'persistent_bf16x3'])
def test_cfg2_size_head_vs_oracle(vkn, chain):
    """(All three forms of the [N x C] chain: the few-row chain — the default at one frame —, one launch per GEMM with the row epilogue
    in the producer, and the persistent row-owner kernels.)
    One 1024x2048 frame through the video head (S=3, N=117, link + x4 upsample) against the CPU oracle.

    At this size (3.8 M logits per stage) a few logits lie within 1e-5 of the binarisation flip point, and ANY change of
    fp32 summation order flips them (SURVEY.md §7 'Threshold semantics'; with i.i.d. random features one flipped pixel moves
    that kernel's x_feat by ~1 %).  So parity is checked TEACHER-FORCED: every stage gets the oracle's previous-stage outputs
    as inputs (identical fp32 mask logits => bit-identical binary masks), and the fused 3-stage call is then checked to be
    bit-identical to the GPU's own stage-by-stage path."""
    case = dict(C=256, heads=8, ffn=2048, ncls=19, n_thing=2, n_stuff=17, S=3, up=4, nprop=100, N=117, H=128, W=256,
                B=1, seed=11, video=1)
    head, (x, pf, mp, prev) = _build_head(vkn, case)
    for h in head.mask_head:
        h.vkn_flags = _chain_flags(vkn, chain)
    traces = []
    obj_r, cls_r, masks_r, scaled_r, track_r = run_oracle(case, traces=traces)
    xd, prevd = _cuda(x, prev)
    obj_in, m_in = pf, mp
    with torch.no_grad():
        for s in range(3):
            kw = dict(previous_obj_feats=prevd) if s == 2 else {}
            r = head._mask_forward(s, xd, obj_in.to(DEV), m_in.to(DEV), [dict()], **kw)
            tr = traces[s]
            cnt_ref = tr['bin_mask'].sum(dim=(-1, -2))
            assert maxabs(r['x_feats'], tr['x_feat']) < 2e-5 * float(tr['x_feat'].abs().max()), f'stage {s} x_feat'
            assert maxabs(r['object_feats'], tr['obj_feat']) < 1e-4, f'stage {s} obj'
            assert maxabs(r['cls_score'], tr['cls_score']) < 1e-4, f'stage {s} cls'
            assert maxabs(r['mask_preds'], tr['new_mask_preds']) < TOL_LOGIT, f'stage {s} mask logits'
            mr = tr['new_mask_preds'].numpy()
            margin = np.abs(mr) > 2e-3
            assert np.array_equal((r['mask_preds'].cpu().numpy() > 0)[margin], (mr > 0)[margin]), f'stage {s} binary masks'
            assert cnt_ref.shape == (1, 117)
            obj_in, m_in = tr['obj_feat'], tr['new_mask_preds']                 # teacher forcing
        assert maxabs(r['object_feats_track'], track_r) < 1e-4
        assert maxabs(r['scaled_mask_preds'], scaled_r) < TOL_LOGIT
        # fused call == the GPU's own stage-by-stage path, bit for bit
        obj, cls, masks, scaled, track = head._head_forward(*_cuda(x, pf, mp, prev), want_track=True)
        o2, m2 = pf.to(DEV), mp.to(DEV)
        for s in range(3):
            kw = dict(previous_obj_feats=prevd) if s == 2 else {}
            r = head._mask_forward(s, xd, o2, m2, [dict()], **kw)
            o2, m2 = r['object_feats'], r['mask_preds']
    assert torch.equal(masks, m2) and torch.equal(obj, o2) and torch.equal(scaled, r['scaled_mask_preds'])
    assert torch.equal(track, r['object_feats_track']) and torch.equal(cls, r['cls_score'].sigmoid())


def test_cfg2_size_batch_of_32_default_policy_vs_oracle(vkn):
    """the bench's ACTUAL path at BASELINE cfg2 size — 32 frames per call, default policy (3744 rows: the persistent
    row-owner chain, the fused decode -> gather pass, the in-call clip link) — against the CPU oracle.  Two distinct frames A / B are
    run through the oracle; the batch holds 16 copies of each (A B A B ...).  Teacher-forced per stage (every stage gets the oracle's
    previous-stage outputs, tiled over the batch): frames 0 and 1 meet the oracle within the parity tolerances, and every other
    frame is BIT-IDENTICAL to its twin (batch invariance: the position of a frame in the batch changes nothing).  Then the fused
    32-frame call is bit-identical to the GPU's own stage-by-stage path."""
    B = 32
    base = dict(C=256, heads=8, ffn=2048, ncls=19, n_thing=2, n_stuff=17, S=3, up=4, nprop=100, N=117, H=128, W=256, B=1, video=1)
    cases = [dict(base, seed=21), dict(base, seed=22)]
    head, _ = _build_head(vkn, cases[0])                 # (weights depend on the seed: both frames run under frame A's weights)
    cfg, sd, *_ = make_case(cases[0])
    ins, traces, tracks = [], [], []
    for c in cases:
        _, _, x, pf, mp, prev = make_case(c)
        tr = []
        with torch.no_grad():
            import oracle.knet_oracle as O
            tracks.append(O.iter_head_mask_preds(sd, x, pf, mp, cfg, previous_obj_feats=prev, traces=tr)[4])
        ins.append((x, pf, mp, prev))
        traces.append(tr)

    def tile(a, b):      # [1, ...] x 2 -> [32, ...]: A B A B ...
        return torch.cat([a, b], 0).repeat(B // 2, *([1] * (a.dim() - 1)))

    xd = tile(ins[0][0], ins[1][0]).to(DEV)
    prevd = tile(ins[0][3], ins[1][3]).to(DEV)
    obj_in, m_in = tile(ins[0][1], ins[1][1]), tile(ins[0][2], ins[1][2])
    metas = [dict()] * B
    with torch.no_grad():
        for s in range(3):
            kw = dict(previous_obj_feats=prevd) if s == 2 else {}
            r = head._mask_forward(s, xd, obj_in.to(DEV), m_in.to(DEV), metas, **kw)
            for f in (0, 1):
                tr = traces[f][s]
                assert maxabs(r['x_feats'][f:f + 1], tr['x_feat']) < 2e-5 * float(tr['x_feat'].abs().max()), f'stage {s} frame {f} x_feat'
                assert maxabs(r['object_feats'][f:f + 1], tr['obj_feat']) < 1e-4, f'stage {s} frame {f} obj'
                assert maxabs(r['cls_score'][f:f + 1], tr['cls_score']) < 1e-4, f'stage {s} frame {f} cls'
                assert maxabs(r['mask_preds'][f:f + 1], tr['new_mask_preds']) < TOL_LOGIT, f'stage {s} frame {f} mask logits'
                mr = tr['new_mask_preds'].numpy()
                margin = np.abs(mr) > 2e-3
                assert np.array_equal((r['mask_preds'][f:f + 1].cpu().numpy() > 0)[margin], (mr > 0)[margin]), f'stage {s} frame {f} binary masks'
            for k in ('x_feats', 'object_feats', 'cls_score', 'mask_preds'):
                t = r[k]
                assert torch.equal(t[0::2], t[0:1].expand_as(t[0::2])) and torch.equal(t[1::2], t[1:2].expand_as(t[1::2])), f'stage {s} {k}: batch invariance'
            obj_in = tile(traces[0][s]['obj_feat'], traces[1][s]['obj_feat'])                 # teacher forcing
            m_in = tile(traces[0][s]['new_mask_preds'], traces[1][s]['new_mask_preds'])
        for f in (0, 1):
            assert maxabs(r['object_feats_track'][f:f + 1], tracks[f]) < 1e-4   # (teacher-forced last stage == the free-running oracle's last stage inputs)
        del r
        torch.cuda.empty_cache()
        # the fused 32-frame call (one C call: persistent chain, fused passes, side-stream link) == the GPU's stage-by-stage path, bit for bit
        x0, pf0, mp0, prev0 = xd, tile(ins[0][1], ins[1][1]).to(DEV), tile(ins[0][2], ins[1][2]).to(DEV), prevd
        obj, cls, masks, scaled, track = head._head_forward(x0, pf0, mp0, prev0, want_track=True)
        o2, m2 = pf0, mp0
        for s in range(3):
            kw = dict(previous_obj_feats=prev0) if s == 2 else {}
            r = head._mask_forward(s, x0, o2, m2, metas, **kw)
            o2, m2 = r['object_feats'], r['mask_preds']
        assert torch.equal(masks, m2) and torch.equal(obj, o2) and torch.equal(scaled, r['scaled_mask_preds'])
        assert torch.equal(track, r['object_feats_track']) and torch.equal(cls, r['cls_score'].sigmoid())
        assert torch.equal(masks[0::2], masks[0:1].expand_as(masks[0::2])) and torch.equal(masks[1::2], masks[1:2].expand_as(masks[1::2]))
    head.check_status()


# ------------------------------------------------------------------------------------------ kernel initialisation ("pass 0")
def _init_rows_off_threshold(masks_golden, nprop, thr, margin=1e-4):
    """rows (b, n) of the thing masks without any logit within `margin` of the binarisation threshold: their object features are
    insensitive to fp32 summation order of the decode."""
    m = torch.from_numpy(masks_golden[:, :nprop])
    return ((m - thr).abs().flatten(2).min(dim=2).values > margin)


@pytest.mark.parametrize('flags', [0, 1], ids=['mfma', 'refkernels'])
@pytest.mark.parametrize('name', ['init_tiny', 'init_odd', 'init_cfg'])
def test_kernel_init_vs_oracle_and_reference(vkn, name, flags):
    """vkn_kernel_init_f32 vs the reference's ConvKernelHead.simple_test_rpn (golden) and vs the oracle."""
    from helpers import load_init_golden, make_init_case
    g, case = load_init_golden(name)
    loc, sem, iw, sw, sb = make_init_case(case)
    dl, ds, diw, dsw, dsb = _cuda(loc, sem, iw, sw, sb)
    prop, xf, masks, seg = vkn.ops.kernel_init(dl, ds, diw, dsw, dsb, case['n_thing'], bool(case['cat']), True, flags=flags)
    torch.cuda.synchronize()
    nprop = case['nprop']
    assert tuple(masks.shape) == g['mask_preds'].shape and tuple(prop.shape) == g['proposal_feats'].shape[:3]
    assert maxabs(masks, g['mask_preds']) < TOL_LOGIT * 0.1
    if seg is not None:
        assert maxabs(seg, g['seg_preds']) < TOL_LOGIT * 0.1
        # the concatenated stuff rows are the seg_preds rows, bit for bit
        assert torch.equal(masks[:, nprop:], seg[:, case['n_thing']:])
    # x_feats = sem + loc: one fp32 add per element -> bit-exact
    if sem is not None:
        assert torch.equal(xf.cpu(), sem + loc)
    else:
        assert xf.data_ptr() == dl.data_ptr()
    # proposal_feats: (a) teacher-forced through the oracle's einsum on the GPU's own logits -> tight
    thr = vkn.ops.thr_logit(0.5)
    bits = (masks[:, :nprop].cpu() >= thr).float()
    obj = torch.einsum('bnhw,bchw->bnc', bits.double(), (xf.cpu()).double())
    want = iw.reshape(1, nprop, -1).double() + obj
    got = prop[:, :nprop].cpu().double()
    scale = float(want.abs().max())
    assert float((got - want).abs().max()) < 2e-5 * max(scale, 1.0)
    # (b) against the reference golden on every row whose logits stay clear of the threshold
    ok = _init_rows_off_threshold(g['mask_preds'], nprop, thr)
    assert float(ok.float().mean()) > 0.3
    gp = torch.from_numpy(g['proposal_feats']).reshape(prop.shape).double()
    err = (prop.cpu().double() - gp)[:, :nprop][ok]
    assert float(err.abs().max()) < 2e-4 * max(scale, 1.0)
    if case['cat']:
        # stuff kernels are copies of conv_seg.weight[num_thing:]
        assert torch.equal(prop[:, nprop:].cpu(), sw[case['n_thing']:].reshape(1, -1, case['C']).expand(case['B'], -1, -1))


def test_kernel_init_soft_weights_vs_reference(vkn):
    """`use_binary=False`: gather weights (sigmoid(z) > 0.5) * sigmoid(z) (knet/det/kernel_head.py:246-247), real-valued left
    operand of k_gather_mfma<., 3>: vs the reference golden (rows clear of the threshold) and teacher-forced vs fp64."""
    from helpers import load_init_golden, make_init_case
    g, case = load_init_golden('init_soft')
    loc, sem, iw, sw, sb = make_init_case(case)
    prop, xf, masks, seg = vkn.ops.kernel_init(*_cuda(loc, sem, iw, sw, sb), case['n_thing'], bool(case['cat']), True, use_binary=False)
    nprop = case['nprop']
    assert maxabs(masks, g['mask_preds']) < TOL_LOGIT * 0.1
    z = masks[:, :nprop].cpu().double()
    w = (z >= vkn.ops.thr_logit(0.5)).double() * torch.sigmoid(masks[:, :nprop].cpu()).double()
    want = iw.reshape(1, nprop, -1).double() + torch.einsum('bnhw,bchw->bnc', w, xf.cpu().double())
    scale = float(want.abs().max())
    assert float((prop[:, :nprop].cpu().double() - want).abs().max()) < 2e-5 * scale
    ok = _init_rows_off_threshold(g['mask_preds'], nprop, vkn.ops.thr_logit(0.5))
    gp = torch.from_numpy(g['proposal_feats']).reshape(prop.shape).double()
    assert float((prop.cpu().double() - gp)[:, :nprop][ok].abs().max()) < 2e-4 * scale
    # and it is NOT the binary gather
    pb = vkn.ops.kernel_init(*_cuda(loc, sem, iw, sw, sb), case['n_thing'], bool(case['cat']), True)[0]
    assert float((pb - prop)[:, :nprop].abs().max()) > 1e-2 * scale


def test_kernel_init_feeds_the_head(vkn):
    """pass 0 -> S-stage head on the GPU equals the oracle chain fed with the same (GPU) pass-0 outputs."""
    from helpers import load_init_golden, make_init_case
    g, icase = load_init_golden('init_tiny')
    loc, sem, iw, sw, sb = make_init_case(icase)
    prop, xf, masks, _ = vkn.ops.kernel_init(*_cuda(loc, sem, iw, sw, sb), icase['n_thing'], True, True)
    _, hcase = load_golden('det_tiny')   # same C / N / H / W as init_tiny
    assert hcase['N'] == masks.shape[1] and hcase['C'] == icase['C']
    cfg, sd, *_ = make_case(hcase)
    head, _ = _build_head(vkn, hcase)
    o, c, m, sc = head.simple_test_mask_preds(xf, prop.reshape(*prop.shape, 1, 1), masks, None, [dict()] * icase['B'])
    with torch.no_grad():
        ro, rc, rm, rsc, _ = O.iter_head_mask_preds(sd, xf.cpu(), prop.cpu().reshape(*prop.shape, 1, 1), masks.cpu(), cfg)
    assert maxabs(m, rm) < TOL_LOGIT and maxabs(c, rc) < 1e-4 and maxabs(o.reshape(ro.shape), ro) < 1e-3


def test_kernel_init_cfg2_size(vkn):
    """BASELINE cfg2 size: shared-kernel decode + strided gather at B = 2, 128x256, against fp64 on the device."""
    B, Np, ncls, nth, C, H, W = 2, 100, 19, 2, 256, 128, 256
    loc, sem = _rand((B, C, H, W), 501).to(DEV), _rand((B, C, H, W), 502).to(DEV)
    iw = (_rand((Np, C, 1, 1), 503, 0.05)).to(DEV)
    sw, sb = _rand((ncls, C, 1, 1), 504, 0.05).to(DEV), _rand((ncls,), 505).to(DEV)
    prop, xf, masks, seg = vkn.ops.kernel_init(loc, sem, iw, sw, sb, nth, True, True)
    assert torch.equal(xf, sem + loc)
    want_m = torch.einsum('nc,bchw->bnhw', iw.reshape(Np, C).double(), loc.double())
    want_s = torch.einsum('nc,bchw->bnhw', sw.reshape(ncls, C).double(), sem.double()) + sb.double().view(1, -1, 1, 1)
    assert maxabs(masks[:, :Np], want_m) < 1e-4 and maxabs(seg, want_s) < 1e-4
    assert torch.equal(masks[:, Np:], seg[:, nth:])
    bits = (masks[:, :Np] >= vkn.ops.thr_logit(0.5)).double()
    want_p = iw.reshape(1, Np, C).double() + torch.einsum('bnhw,bchw->bnc', bits, xf.double())
    assert maxabs(prop[:, :Np], want_p) < 2e-5 * float(want_p.abs().max())


@pytest.mark.parametrize('B,Np,ncls,nth,C,H,W,cat,seg', [(2, 100, 19, 2, 256, 128, 256, True, True), (3, 100, 19, 2, 256, 16, 32, True, False),
                                                          (1, 127, 1, 0, 256, 8, 16, False, True), (2, 97, 31, 11, 128, 8, 24, True, True),
                                                          (2, 12, 5, 2, 64, 8, 16, True, True), (1, 20, 12, 3, 256, 16, 16, False, False)])
def test_kernel_init_one_pass_equals_the_separate_form_bit_for_bit(vkn, B, Np, ncls, nth, C, H, W, cat, seg):
    """Round 6: pass 0 as ONE pass over loc and sem (k_init_pass: both 1x1 decodes, x = loc + sem, the stuff rows, the thing bits of
    the kernel-init gather; knet/det/kernel_head.py:222-257) against the round-5 form (two decode launches, a copy, an add pass, a
    logits gather; VKN_FLAG_INIT_SEPARATE) — which the reference goldens pin: all four outputs bit for bit, over both built row maps
    (100 proposals + 19 classes in four n-blocks; everything in one), ragged row counts, with / without stuff rows and seg_preds."""
    assert vkn._lib.lib().vkn_kernel_init_workspace_bytes(B, Np, ncls, C, H * W) > 0
    loc, sem = _rand((B, C, H, W), 511).to(DEV), _rand((B, C, H, W), 512).to(DEV)
    iw = (_rand((Np, C, 1, 1), 513, 0.05)).to(DEV)
    sw, sb = _rand((ncls, C, 1, 1), 514, 0.05).to(DEV), _rand((ncls,), 515).to(DEV)
    a = vkn.ops.kernel_init(loc, sem, iw, sw, sb, nth, cat, True, want_seg_preds=seg)
    b = vkn.ops.kernel_init(loc, sem, iw, sw, sb, nth, cat, True, want_seg_preds=seg, flags=vkn.ops.FLAG_INIT_SEPARATE)
    for nm, u, v in zip(('proposal_feats', 'x_feats', 'mask_preds', 'seg_preds'), a, b):
        assert (u is None and v is None) or torch.equal(u, v), nm
    assert torch.equal(a[1], sem + loc)
    c = vkn.ops.kernel_init(loc, sem, iw, sw, sb, nth, cat, False, want_seg_preds=seg)         # no object features: no bits, no gather
    assert torch.equal(c[2], a[2]) and torch.equal(c[0][:, :Np], iw.reshape(1, Np, C).expand(B, -1, -1))


def test_conv_kernel_head_class_matches_reference_golden(vkn):
    """The registry class end to end: a pass-through neck, reference-shaped checkpoint, `simple_test_rpn` 5-tuple."""
    from helpers import load_init_golden, make_init_case
    g, case = load_init_golden('init_cfg')
    loc, sem, iw, sw, sb = make_init_case(case)

    class Neck(torch.nn.Module):
        def forward(self, feats):
            return [feats[0], feats[1]]

    head = vkn.build_head(dict(type='ConvKernelHead', num_proposals=case['nprop'], in_channels=case['C'], out_channels=case['C'],
                               num_loc_convs=0, num_seg_convs=0, localization_fpn=Neck(), semantic_fpn=True,
                               num_classes=case['ncls'], use_binary=True, proposal_feats_with_obj=True,
                               num_thing_classes=case['n_thing'], num_stuff_classes=case['ncls'] - case['n_thing'],
                               cat_stuff_mask=True))
    head.load_state_dict({'init_kernels.weight': iw, 'conv_seg.weight': sw, 'conv_seg.bias': sb}, strict=True)
    head = head.to(DEV).eval()
    prop, xf, masks, cls, seg = head.simple_test_rpn((loc.to(DEV), sem.to(DEV)), [dict()] * case['B'])
    assert cls is None and tuple(prop.shape) == g['proposal_feats'].shape
    assert maxabs(masks, g['mask_preds']) < TOL_LOGIT * 0.1 and maxabs(seg, g['seg_preds']) < TOL_LOGIT * 0.1
    ok = _init_rows_off_threshold(g['mask_preds'], case['nprop'], vkn.ops.thr_logit(0.5))
    err = (prop.cpu() - torch.from_numpy(g['proposal_feats']))[:, :case['nprop']].flatten(2)[ok]
    assert float(err.abs().max()) < 2e-4 * float(np.abs(g['proposal_feats']).max())


# --------------------------------------------------------------------------------------- post-head pipeline: joint panoptic merge
def _pan_gpu(vkn, case):
    from helpers import PAN_CFG, make_pan_case
    cls, logits, meta = make_pan_case(case)
    seg, info, nseg = vkn.ops.panoptic_joint(cls.to(DEV), logits.to(DEV), case['Np'], case['T'], case['Np'],
                                             PAN_CFG['instance_score_thr'], PAN_CFG['overlap_thr'], meta['img_shape'][:2],
                                             meta['batch_input_shape'], meta['ori_shape'][:2], upsample_stride=case['up'])
    torch.cuda.synchronize()
    return seg.cpu().numpy(), info.cpu().numpy(), nseg.cpu().numpy()


@pytest.mark.parametrize('name', ['pan_tiny', 'pan_ident', 'pan_cfg', 'pan_kitti', 'pan_vipseg'])
def test_panoptic_joint_vs_oracle_and_reference(vkn, name):
    """Integer artefacts of the post-head pipeline: selection (rows / labels / scores) and segment decisions bit-exact; the
    panoptic map bit-exact except where the reference's own arg-max is decided by < 1e-6 (fp32 resampling noise)."""
    from helpers import load_pan_golden, run_pan_oracle
    g, case = load_pan_golden(name)
    seg, info, nseg = _pan_gpu(vkn, case)
    for b in range(case['B']):
        r = run_pan_oracle(case, b)
        assert int(nseg[b]) == len(r['segments_info']) == int(g['nseg'][b])
        assert np.array_equal(info[b, :, 0], r['rows'].numpy()) and np.array_equal(info[b, :, 1], r['total_labels'].numpy())
        assert np.array_equal(info[b, :, 5].view(np.float32), r['total_scores'].numpy())       # score bits
        assert np.array_equal(info[b, :, 2], r['seg_of'].numpy())                              # accept / reject + ids
        near = (r['margin'].numpy() < 1e-6)
        assert float(near.mean()) < 2e-3
        n_near = int(near.sum())
        assert np.abs(info[b, :, 3] - r['area'].numpy()).sum() <= 2 * n_near                   # pixels won
        near_half = int(((r['total_masks'] - 0.5).abs() < 1e-6).sum())
        assert np.abs(info[b, :, 4] - r['orig'].numpy()).sum() <= near_half                    # pixels with prob >= 0.5
        diff = seg[b] != g['panoptic_seg'][b]
        assert not (diff & ~near).any()
        assert int(diff.sum()) <= n_near


def test_panoptic_joint_cfg2_size(vkn):
    """BASELINE cfg2 geometry (128x256 logits, x4, 1024x2048 output, K = 117): consistency of the integer outputs and
    agreement with a straightforward torch evaluation of the same chain on the device (which materialises K x Ho x Wo)."""
    B, N, Np, T, ncls, Hm, Wm, up = 1, 117, 100, 2, 19, 128, 256, 4
    cls_np, logit_np = synth.panoptic_inputs(B, N, Np, ncls, Hm, Wm, 77)
    cls, logits = torch.from_numpy(cls_np).to(DEV), torch.from_numpy(logit_np).to(DEV)
    shape = (1024, 2048)
    seg, info, nseg = vkn.ops.panoptic_joint(cls, logits, Np, T, Np, 0.25, 0.6, shape, shape, shape, upsample_stride=up)
    torch.cuda.synchronize()
    info = info[0].cpu().numpy()
    K = info.shape[0]
    assert int(nseg[0]) > 5 and int(nseg[0]) == int((info[:, 2] > 0).sum())
    assert int(info[:, 3].sum()) == shape[0] * shape[1]                      # every pixel is won by exactly one kernel
    # torch chain on the device
    rows = torch.from_numpy(info[:, 0]).long().to(DEV)
    scores = torch.from_numpy(info[:, 5].view(np.float32).copy()).to(DEV)
    scaled = F.interpolate(logits[0][None], scale_factor=up, mode='bilinear', align_corners=False)[0]
    tm = F.interpolate(scaled[rows][None].sigmoid(), size=shape, mode='bilinear', align_corners=False)[0]
    prob = scores.view(-1, 1, 1) * tm
    top2 = prob.topk(2, dim=0)
    near = (top2.values[0] - top2.values[1]) < 1e-6
    ids = top2.indices[0]
    seg_of = torch.from_numpy(info[:, 2]).to(DEV)
    want = seg_of[ids].int()
    diff = seg[0] != want
    assert not bool((diff & ~near).any()) and float(near.float().mean()) < 2e-3
    area = torch.bincount(ids.flatten(), minlength=K).cpu().numpy()
    assert np.abs(area - info[:, 3]).sum() <= 2 * int(near.sum())
    orig = (tm >= 0.5).flatten(1).sum(1).cpu().numpy()
    assert np.abs(orig - info[:, 4]).sum() <= int(((tm - 0.5).abs() < 1e-6).sum())
    # determinism
    seg2, info2, _ = vkn.ops.panoptic_joint(cls, logits, Np, T, Np, 0.25, 0.6, shape, shape, shape, upsample_stride=up)
    assert torch.equal(seg, seg2) and np.array_equal(info, info2[0].cpu().numpy())


_TEST_CFG = dict(max_per_img=12, mask_thr=0.5, stuff_score_thr=0.05,
                 merge_stuff_thing=dict(overlap_thr=0.6, iou_thr=0.5, stuff_max_area=4096, instance_score_thr=0.25))


def _pan_compare(seg, segments_info, r):
    """GPU (panoptic_seg ndarray, segments_info) vs an oracle.panoptic_joint result, modulo arg-max near-ties."""
    from helpers import pan_info_rows
    near = r['margin'].numpy() < 1e-6
    diff = seg != r['panoptic_seg'].numpy()
    assert not (diff & ~near).any() and float(near.mean()) < 5e-3
    a, b = pan_info_rows(segments_info), pan_info_rows(r['segments_info'])
    assert a.shape == b.shape
    assert np.array_equal(np.nan_to_num(a[:, :5], nan=-7.0), np.nan_to_num(b[:, :5], nan=-7.0))      # ids, classes, scores
    assert np.abs(a[:, 5] - b[:, 5]).sum() <= 2 * int(near.sum())                                        # stuff areas


def test_simple_test_panoptic_path(vkn):
    """KernelIterHead.simple_test (do_panoptic, merge_joint): stage loop + fused post-head pipeline from the low-res logits.
    Checked against the oracle's panoptic_joint fed with the GPU head's own cls / mask outputs, and against the reference's
    route through the materialised `scaled_mask_preds` (get_panoptic signature)."""
    from test_host_logic import _cfg
    _, case = load_golden('det_tiny')
    cfg = _cfg(False, C=case['C'], heads=case['heads'], ffn=case['ffn'], ncls=case['ncls'], n_thing=case['n_thing'],
               n_stuff=case['n_stuff'], S=case['S'], up=case['up'], nprop=case['nprop'])
    cfg.update(do_panoptic=True, merge_joint=True, test_cfg=_TEST_CFG)
    head = vkn.build_head(cfg)
    _, sd, x, pf, mp, _ = make_case(case)
    head.load_state_dict(sd, strict=True)
    head = head.to(DEV).eval()
    H, W, up = case['H'], case['W'], case['up']
    meta = dict(img_shape=(H * 8 - 3, W * 8 - 5, 3), batch_input_shape=(H * 8, W * 8), ori_shape=(H * 6, W * 6, 3))
    metas = [meta] * case['B']
    dx, dpf, dmp = _cuda(x, pf, mp)
    res = head.simple_test(dx, dpf, dmp, None, metas)
    o, c, m, sc = head.simple_test_mask_preds(dx, dpf, dmp, None, metas)
    assert len(res) == case['B']
    for b in range(case['B']):
        bbox, segm, (seg, info) = res[b]
        assert bbox is None and segm is None and seg.dtype == np.int32 and seg.shape == (H * 6, W * 6)
        with torch.no_grad():
            r = O.panoptic_joint(c[b].cpu(), m[b].cpu(), case['nprop'], case['n_thing'], 12, 0.25, 0.6, meta, upsample_stride=up)
        _pan_compare(seg, info, r)
        # the reference's own route: get_panoptic on the materialised scaled masks
        _, _, (seg2, info2) = head.get_panoptic(c[b], sc[b], head.test_cfg, meta)
        _pan_compare(seg2, info2, r)


def test_video_simple_test_with_previous(vkn):
    """VideoKernelIterHead.simple_test_with_previous: results + tracking embeddings of the accepted thing segments."""
    from test_host_logic import _cfg
    _, case = load_golden('video_tiny')
    cfg = _cfg(True, C=case['C'], heads=case['heads'], ffn=case['ffn'], ncls=case['ncls'], n_thing=case['n_thing'],
               n_stuff=case['n_stuff'], S=case['S'], up=case['up'], nprop=case['nprop'])
    cfg.update(do_panoptic=True, merge_joint=True, with_track=True, test_cfg=_TEST_CFG)
    head = vkn.build_head(cfg)
    _, sd, x, pf, mp, prev = make_case(case)
    head.load_state_dict(sd, strict=True)
    head = head.to(DEV).eval()
    H, W = case['H'], case['W']
    meta = dict(img_shape=(H * 8, W * 8, 3), batch_input_shape=(H * 8, W * 8), ori_shape=(H * 8, W * 8, 3))
    metas = [meta] * case['B']
    dx, dpf, dmp, dprev = _cuda(x, pf, mp, prev)
    results, obj, cls, masks, scaled = head.simple_test_with_previous(dx, dpf, dmp, None, metas, previous_obj_feats=dprev)
    o2, c2, m2, s2, track = head.simple_test_mask_preds_plus_previous(dx, dpf, dmp, None, metas, previous_obj_feats=dprev,
                                                                      return_track=True)
    assert torch.equal(masks, m2) and torch.equal(cls, c2) and torch.equal(scaled, s2)
    for b in range(case['B']):
        bbox, segm, tmask, (seg, info), tfeat = results[b]
        with torch.no_grad():
            r = O.panoptic_joint(cls[b].cpu(), masks[b].cpu(), case['nprop'], case['n_thing'], 12, 0.25, 0.6, meta,
                                 upsample_stride=case['up'])
        _pan_compare(seg, info, r)
        things = [s for s in info if s['isthing']]
        assert tfeat.shape[0] == len(things)
        rows = r['rows'].numpy()
        for j, s_ in enumerate(things):
            assert torch.equal(tfeat[j], track[b, int(rows[s_['instance_id']])])


@pytest.mark.parametrize('name', ['pan_tf_tiny', 'pan_tf_cfg'])
def test_thing_first_merge_vs_reference_golden(vkn, name):
    """merge_joint=False: `get_panoptic` -> `merge_stuff_thing` (knet/det/kernel_iter_head.py:385-465) on the device: segment list
    (ids, kinds, labels, instance indices, scores) equal, stuff areas and the map equal up to the few pixels whose rescaled
    probability sits on the 0.5 threshold."""
    from helpers import load_pan_golden, make_pan_case, pan_info_rows
    g, case = load_pan_golden(name)
    cls, logits, meta = make_pan_case(case)
    cfg = vkn.configs.roi_head_cfg(False, C=32, heads=8, ffn=64, ncls=case['ncls'], n_thing=case['T'], n_stuff=case['ncls'] - case['T'],
                                   S=1, up=case['up'], nprop=case['Np'], merge_joint=False)
    cfg['test_cfg'] = dict(max_per_img=case['Np'], mask_thr=0.5,
                           merge_stuff_thing=dict(overlap_thr=0.6, iou_thr=0.5, stuff_max_area=int(g['stuff_max_area']), instance_score_thr=0.25))
    head = vkn.build_head(cfg).to(DEV).eval()
    scaled = F.interpolate(logits, scale_factor=case['up'], align_corners=False, mode='bilinear') if case['up'] > 1 else logits
    for b in range(case['B']):
        bbox_result, segm_result, (seg, info) = head.get_panoptic(cls[b].to(DEV), scaled[b].to(DEV), head.test_cfg, meta)
        rows, ref = pan_info_rows(info), g[f'info{b}']
        assert rows.shape == ref.shape and np.array_equal(rows[:, :4], ref[:, :4])
        assert np.allclose(rows[:, 4], ref[:, 4], rtol=0, atol=1e-6, equal_nan=True)
        stuff = ref[:, 1] == 0
        assert np.all(np.abs(rows[stuff, 5] - ref[stuff, 5]) <= 0.002 * ref[stuff, 5] + 8)
        assert np.mean(seg != g['panoptic_seg'][b]) < 2e-3
        assert sum(len(m) for m in segm_result) == int(g[f'nmask{b}'])


def test_instance_only_results_vs_reference_golden(vkn):
    """do_panoptic=False (BASELINE cfg4's result path): `simple_test` -> per image (bbox_result, segm_result) — scores, labels and
    the full-resolution boolean masks against the reference's own simple_test (knet/det/kernel_iter_head.py:270-281)."""
    g, case = load_golden('inst_tiny')
    cfg = vkn.configs.roi_head_cfg(False, C=case['C'], heads=case['heads'], ffn=case['ffn'], ncls=case['ncls'], n_thing=case['n_thing'],
                                   n_stuff=case['n_stuff'], S=case['S'], up=case['up'], nprop=case['nprop'], do_panoptic=False,
                                   test_cfg=dict(max_per_img=10, mask_thr=0.5))
    head = vkn.build_head(cfg)
    _, sd, x, pf, mp, _ = make_case(case)
    head.load_state_dict(sd, strict=True)
    head = head.to(DEV).eval()
    meta = dict(img_shape=(60, 120, 3), batch_input_shape=(64, 128), ori_shape=(90, 180, 3))
    with torch.no_grad():
        res = head.simple_test(*_cuda(x, pf, mp), None, [meta] * case['B'])
    for b, (bbox_result, segm_result) in enumerate(res):
        assert len(bbox_result) == case['ncls'] and len(segm_result) == case['ncls']
        scores = np.concatenate([bb[:, 4] for bb in bbox_result])
        labels = np.concatenate([np.full(len(bb), c) for c, bb in enumerate(bbox_result)])
        assert np.array_equal(labels, g[f'labels{b}']) and np.max(np.abs(scores - g[f'scores{b}'])) < 1e-5
        masks = np.stack([m for per_cls in segm_result for m in per_cls]).astype(bool)
        ref = np.unpackbits(g[f'masks{b}'])[:masks.size].reshape(masks.shape).astype(bool)
        assert masks.shape == (int(g[f'nmask{b}']), 90, 180) and np.mean(masks != ref) < 1e-3


def test_video_get_panoptic_vs_reference_golden(vkn):
    """VideoKernelIterHead.get_panoptic (knet/video/kernel_iter_head.py:591-640): panoptic map, segments and `thing_obj_feat` —
    the tracking embeddings of the accepted things in segment order — against the reference's own video head."""
    from helpers import load_pan_golden, make_pan_case, pan_info_rows
    g, case = load_pan_golden('pan_video')
    cls, logits, meta = make_pan_case(case)
    obj = torch.from_numpy(synth.normalish((case['B'], case['N'], 32), 77 + case['seed'], 1.0))
    cfg = vkn.configs.roi_head_cfg(True, C=32, heads=8, ffn=64, ncls=case['ncls'], n_thing=case['T'], n_stuff=case['ncls'] - case['T'],
                                   S=1, up=case['up'], nprop=case['Np'])
    cfg['test_cfg'] = dict(max_per_img=case['Np'], mask_thr=0.5, merge_stuff_thing=dict(overlap_thr=0.6, instance_score_thr=0.25))
    head = vkn.build_head(cfg).to(DEV).eval()
    scaled = F.interpolate(logits, scale_factor=case['up'], align_corners=False, mode='bilinear') if case['up'] > 1 else logits
    for b in range(case['B']):
        bbox, segm, tmask, (seg, info), tfeat = head.get_panoptic(cls[b].to(DEV), scaled[b].to(DEV), head.test_cfg, meta,
                                                                   obj_feat=obj[b].to(DEV))
        rows = pan_info_rows(info)
        ref = g[f'info{b}']
        assert rows.shape == ref.shape and np.array_equal(rows[:, :4], ref[:, :4])
        assert np.allclose(rows[:, 4], ref[:, 4], rtol=0, atol=1e-6, equal_nan=True)
        assert np.mean(seg != g['panoptic_seg'][b]) < 2e-3          # arg-max ties under fp32 resampling noise only
        assert torch.equal(tfeat.cpu(), torch.from_numpy(g[f'thing_obj_feat{b}'])), 'thing_obj_feat = obj_feat[things], exact'


@pytest.mark.parametrize('name', ['det_cfg', 'video_cfg', 'det_tiny'])
def test_bit_packed_stage_handoff_is_exact(vkn, name):
    """The fused head hands stage s -> s+1 the binarised masks as bit words (decode's bit-packed epilogue -> gather's bit
    operand) when H*W % 64 == 0.  VKN_FLAG_LOGITS_HANDOFF (4) keeps the fp32 logits path: every output must be bit-identical."""
    g, case = load_golden(name)
    head, (x, pf, mp, prev) = _build_head(vkn, case)
    assert (case['H'] * case['W']) % 64 == 0
    B, N, C = case['B'], case['N'], case['C']
    dims = head.mask_head[0].make_dims(B, N, case['H'], case['W'])
    packs = [h.stage_pack(torch.device(DEV)) for h in head.mask_head]
    dx, dpf, dmp = _cuda(x, pf.reshape(B, N, C), mp)
    a = vkn.ops.head_forward(dims, packs, dx, dpf, dmp, None, case['up'], flags=0)     # fused decode -> gather pass (C in {64,128,256})
    b = vkn.ops.head_forward(dims, packs, dx, dpf, dmp, None, case['up'], flags=4)     # fp32 logits hand-off
    c = vkn.ops.head_forward(dims, packs, dx, dpf, dmp, None, case['up'], flags=16)    # bit words through two kernels
    torch.cuda.synchronize()
    for u, v, w in zip(a[:4], b[:4], c[:4]):
        assert torch.equal(u, v) and torch.equal(u, w)
    assert maxabs(a[2], g['mask_preds']) < TOL_LOGIT


@pytest.mark.parametrize('hw', [(24, 40), (23, 40)], ids=['bits_handoff', 'logits_handoff'])
def test_head_vipseg_kernel_count_vs_oracle(vkn, hw):
    """VIP-Seg-like dims (N = 166 kernels -> two n-chunks of 128 + 64 rows, 124 classes), S = 2: the fused head against the oracle,
    with the bit-packed hand-off (H*W % 64 == 0) and with the fp32 logits hand-off (H*W = 920)."""
    from test_host_logic import _cfg
    H, W = hw
    case = dict(C=64, heads=8, ffn=128, ncls=124, n_thing=58, n_stuff=66, S=2, up=2, nprop=100, N=166, H=H, W=W, B=2, seed=41,
                video=0)
    head = vkn.build_head(_cfg(False, C=64, heads=8, ffn=128, ncls=124, n_thing=58, n_stuff=66, S=2, up=2, nprop=100))
    cfg, sd, x, pf, mp, _ = make_case(case)
    head.load_state_dict(sd, strict=True)
    head = head.to(DEV).eval()
    o, c, m, sc = head.simple_test_mask_preds(*_cuda(x, pf, mp), None, [dict()] * 2)
    with torch.no_grad():
        traces = []
        ro, rc, rm, rsc, _ = O.iter_head_mask_preds(sd, x, pf, mp, cfg, traces=traces)
    # the stage-0 masks decide stage 1's gather: compare where the oracle's stage-0 logits are not within 1e-4 of the threshold
    assert maxabs(c, rc) < 1e-3 and maxabs(m, rm) < 5e-2   # loose end-to-end bound (a flipped pixel moves a kernel slightly)
    dims = head.mask_head[0].make_dims(2, 166, H, W)
    packs = [h.stage_pack(torch.device(DEV)) for h in head.mask_head]
    a = vkn.ops.head_forward(dims, packs, *_cuda(x, pf.reshape(2, 166, 64), mp), None, 2, flags=0)
    b = vkn.ops.head_forward(dims, packs, *_cuda(x, pf.reshape(2, 166, 64), mp), None, 2, flags=4)
    c = vkn.ops.head_forward(dims, packs, *_cuda(x, pf.reshape(2, 166, 64), mp), None, 2, flags=16)
    for u, v, w in zip(a[:4], b[:4], c[:4]):
        assert torch.equal(u, v) and torch.equal(u, w)
    # teacher-forced last stage: feed the oracle's stage-0 outputs to the GPU stage 1 -> tight
    t0 = traces[0]
    cls1, m1, o1, _, _ = vkn.ops.stage_forward(dims, packs[1], x.to(DEV), t0['obj_feat'].reshape(2, 166, 64).to(DEV),
                                                t0['new_mask_preds'].to(DEV))
    assert maxabs(m1, rm) < TOL_LOGIT and maxabs(cls1, traces[1]['cls_score']) < 1e-4


@pytest.mark.parametrize('C,heads,N', [(256, 4, 40), (256, 4, 166), (256, 8, 20), (256, 8, 60), (256, 8, 230), (128, 8, 117),
                                       (256, 16, 100), (128, 2, 33)])
def test_stage_attention_head_widths_and_key_blocks_vs_oracle(vkn, C, heads, N):
    """The kernel-to-kernel attention on the matrix cores (k_attn_mfma) for every head width it takes (16 / 32 / 64 channels) and
    every key-block count (N = 20 .. 230 kernels -> 1, 2, 4, 6, 8 blocks of 32 keys): one stage, same inputs, against the oracle."""
    from test_host_logic import _cfg
    H, W = 8, 16
    kw = dict(C=C, heads=heads, ffn=2 * C, ncls=19, n_thing=8, n_stuff=11, S=1, up=1, nprop=N - 11)
    case = dict(kw, N=N, H=H, W=W, B=2, seed=300 + N + heads, video=0)
    head = vkn.build_head(_cfg(False, **kw))
    cfg, sd, x, pf, mp, _ = make_case(case)
    head.load_state_dict(sd, strict=True)
    head = head.to(DEV).eval()
    with torch.no_grad():
        traces = []
        O.iter_head_mask_preds(sd, x, pf, mp, cfg, traces=traces)
    dims = head.mask_head[0].make_dims(2, N, H, W)
    pack = head.mask_head[0].stage_pack(torch.device(DEV))
    cls0, m0, o0, _, _ = vkn.ops.stage_forward(dims, pack, x.to(DEV), pf.reshape(2, N, C).to(DEV), mp.to(DEV))
    t0 = traces[0]
    assert maxabs(o0, t0['obj_feat'].reshape(2, N, C)) < 2e-4
    assert maxabs(m0, t0['new_mask_preds']) < TOL_LOGIT and maxabs(cls0, t0['cls_score']) < 1e-4


@pytest.mark.parametrize('xscale,oscale', [(1e-4, 1.0), (1.0, 1e-3), (3e3, 1.0), (40.0, 50.0), (3e3, 1e-3)])
def test_two_term_chain_range_management(vkn, xscale, oscale):
    """The persistent chain on the two-term fp16 split (the default of that form) over inputs that leave fp16's exponent range in both
    directions: update features (gather sums) of magnitude 1e-4 .. 1e5 and incoming kernels of 1e-3 .. 50 — every row of the four
    unbounded activation images is scaled by its own power of two and every weight image by its matrix', so the results stay at fp32
    rounding from the bf16 x 3 form and the exact-fp32 chain (no overflow at 65504, no lost low halves below 2^-14)."""
    from test_host_logic import _cfg
    B, N, C, ff, ncls = 3, 117, 256, 2048, 19
    kw = dict(C=C, heads=8, ffn=ff, ncls=ncls, n_thing=2, n_stuff=17, S=1, up=1, nprop=100)
    head = vkn.build_head(_cfg(False, **kw))
    cfg, sd, x, pf, mp, prev = make_case(dict(kw, N=N, H=8, W=16, B=B, seed=777, video=0))
    head.load_state_dict(sd, strict=True)
    head = head.to(DEV).eval()
    dims = head.mask_head[0].make_dims(B, N, 8, 16)
    pack = head.mask_head[0].stage_pack(torch.device(DEV))
    g = torch.Generator(device='cpu').manual_seed(31)
    xf = (torch.randn(B, N, C, generator=g) * 30 * xscale).to(DEV)
    xf[0, 3] *= 1e-3                                   # rows of very different magnitude inside one tile
    xf[1, 5] = 0.0                                     # an all-zero row (an empty mask)
    ob = (torch.randn(B, N, C, generator=g) * oscale).to(DEV)
    o = vkn.ops
    h2 = o.stage_chain(dims, pack, xf, ob, flags=o.FLAG_CHAIN_PERSISTENT)
    b3 = o.stage_chain(dims, pack, xf, ob, flags=o.FLAG_CHAIN_PERSISTENT | o.FLAG_CHAIN_BF16X3)
    ex = o.stage_chain(dims, pack, xf, ob, flags=o.FLAG_EXACT_GEMM)
    for nm, a, b, e in zip(('cls', 'kernels', 'bias', 'obj'), h2, b3, ex):
        assert torch.isfinite(a).all(), nm
        scale = max(1.0, float(e.abs().max()))
        assert maxabs(a, e) < 4e-5 * scale and maxabs(a, b) < 4e-5 * scale, (nm, maxabs(a, e), maxabs(a, b), maxabs(b, e), scale)
    assert not torch.equal(h2[3], b3[3]), 'the flag must select a different arithmetic'


def test_two_term_chain_reports_activations_outside_the_fp16_envelope(vkn):
    """the persistent chain's UNSCALED activation images (LayerNorm outputs, FFN hidden rows, branch inputs) are O(1) for
    ordinary weights; a fine-tuned / badly scaled W1 or LayerNorm gain that pushes one beyond 2^15 used to become inf -> NaN silently.
    Now the write ORs VKN_STATUS_RANGE into the workspace status word (`workspace_status()` raises VKN_E_RANGE); the bf16x3 form
    (VKN_FLAG_CHAIN_BF16X3: fp32 range) computes the same stage cleanly and is the documented fallback."""
    from test_host_logic import _cfg
    B, N, C, ff, ncls = 3, 117, 256, 2048, 19
    kw = dict(C=C, heads=8, ffn=ff, ncls=ncls, n_thing=2, n_stuff=17, S=1, up=1, nprop=100)
    cfg, sd, x, pf, mp, prev = make_case(dict(kw, N=N, H=8, W=16, B=B, seed=778, video=0))
    g = torch.Generator(device='cpu').manual_seed(32)
    xf = (torch.randn(B, N, C, generator=g) * 30).to(DEV)
    ob = torch.randn(B, N, C, generator=g).to(DEV)
    o = vkn.ops

    def run(scale_key, factor, flags):
        head = vkn.build_head(_cfg(False, **kw))
        sd2 = {k: (v * factor if k == scale_key else v) for k, v in sd.items()}
        head.load_state_dict(sd2, strict=True)
        head = head.to(DEV).eval()
        dims = head.mask_head[0].make_dims(B, N, 8, 16)
        return o.stage_chain(dims, head.mask_head[0].stage_pack(torch.device(DEV)), xf, ob, flags=flags)

    o.workspace_status()                                                   # clear whatever an earlier test left
    out = run(None, 1.0, o.FLAG_CHAIN_PERSISTENT)
    o.workspace_status()                                                   # ordinary weights: nothing reported
    assert all(torch.isfinite(t).all() for t in out)
    for key, factor in (('mask_head.0.ffn.layers.0.0.weight', 3e5), ('mask_head.0.attention_norm.weight', 1e5)):
        run(key, factor, o.FLAG_CHAIN_PERSISTENT)
        with pytest.raises(vkn._lib.VknError) as e:
            o.workspace_status()
        assert e.value.code == -6, key
        o.workspace_status()                                               # read-and-clear
        b3 = run(key, factor, o.FLAG_CHAIN_PERSISTENT | o.FLAG_CHAIN_BF16X3)
        o.workspace_status()                                               # the fp32-range form does not report ...
        ex = run(key, factor, o.FLAG_EXACT_GEMM)
        assert all(torch.isfinite(t).all() for t in b3)                    # ... and computes the stage
        assert maxabs(b3[3], ex[3]) < 1e-4 * max(1.0, float(ex[3].abs().max()))


@pytest.mark.parametrize('B,N,ff,ncls,video', [(1, 117, 2048, 19, 0), (3, 117, 2048, 19, 1), (2, 166, 1024, 124, 0), (5, 20, 512, 40, 0),
                                                (1, 32, 256, 3, 0), (8, 100, 2048, 40, 1), (18, 117, 2048, 19, 1), (5, 117, 2048, 19, 0)])
def test_persistent_chain_equals_launch_per_gemm_chain(vkn, B, N, ff, ncls, video):
    """The [N x C] chain as two persistent row-owner kernels (k_chain_a / k_chain_c, the default at C = 256) against the same chain as
    one launch per GEMM (VKN_FLAG_CHAIN_LAUNCHES) and against the exact-fp32 GEMM chain (VKN_FLAG_EXACT_GEMM): one stage, same
    inputs — ragged last row tile (B*N % 32 != 0), one / many workgroups, 1 .. 8 hidden chunks, class counts below and above one
    column block, with and without the video link.  Same bf16x3 arithmetic, different summation order: fp32 rounding apart."""
    from test_host_logic import _cfg
    C, heads, H, W = 256, 8, 8, 16
    n_stuff = 11 if (N > 11 and ncls > 11) else 1
    kw = dict(C=C, heads=heads, ffn=ff, ncls=ncls, n_thing=ncls - n_stuff, n_stuff=n_stuff, S=1, up=1, nprop=N - n_stuff)
    case = dict(kw, N=N, H=H, W=W, B=B, seed=700 + B + N, video=video)
    head = vkn.build_head(_cfg(bool(video), **kw))
    cfg, sd, x, pf, mp, prev = make_case(case)
    head.load_state_dict(sd, strict=True)
    head = head.to(DEV).eval()
    dims = head.mask_head[0].make_dims(B, N, H, W)
    pack = head.mask_head[0].stage_pack(torch.device(DEV))
    args = (dims, pack, x.to(DEV), pf.reshape(B, N, C).to(DEV), mp.to(DEV))
    kwd = dict(prev_obj=prev.reshape(B, N, C).to(DEV), want_track=True) if video else {}
    new = vkn.ops.stage_forward(*args, flags=vkn.ops.FLAG_CHAIN_PERSISTENT, **kwd)
    old = vkn.ops.stage_forward(*args, flags=vkn.ops.FLAG_CHAIN_LAUNCHES, **kwd)
    few = vkn.ops.stage_forward(*args, flags=vkn.ops.FLAG_CHAIN_KSPLIT, **kwd)      # the few-row chain (vkn_ksplit.hip), forced at any row count
    exact = vkn.ops.stage_forward(*args, flags=vkn.ops.FLAG_EXACT_GEMM, **kwd)
    names = ('cls', 'masks', 'obj', 'x_feat', 'track')
    for nm, a, b, f, e in zip(names, new, old, few, exact):
        if a is None:
            assert b is None and f is None
            continue
        scale = max(1.0, float(e.abs().max()))
        d_old, d_exact, d_base = maxabs(a, b), maxabs(a, e), maxabs(b, e)
        assert torch.isfinite(a).all() and torch.isfinite(f).all(), nm
        # the bf16x3 chains sit equally close to the exact-fp32 chain (all ~1e-6 on O(1) values)
        assert d_old < 2e-5 * scale and d_exact < 2e-5 * scale, (nm, d_old, d_exact, d_base, scale)
        assert maxabs(f, b) < 2e-5 * scale and maxabs(f, e) < 2e-5 * scale, (nm, maxabs(f, b), maxabs(f, e), scale)
    with torch.no_grad():
        traces = []
        O.iter_head_mask_preds(sd, x, pf, mp, cfg, traces=traces, **(dict(previous_obj_feats=prev) if video else {}))
    t0 = traces[0]
    assert maxabs(new[2], t0['obj_feat'].reshape(B, N, C)) < 2e-4
    assert maxabs(new[1], t0['new_mask_preds']) < TOL_LOGIT and maxabs(new[0], t0['cls_score']) < 1e-4
    again = vkn.ops.stage_forward(*args, flags=vkn.ops.FLAG_CHAIN_PERSISTENT, **kwd)
    assert all(a is None or torch.equal(a, b) for a, b in zip(new, again)), 'deterministic'
    again = vkn.ops.stage_forward(*args, flags=vkn.ops.FLAG_CHAIN_KSPLIT, **kwd)
    assert all(a is None or torch.equal(a, b) for a, b in zip(few, again)), 'deterministic (few-row chain)'
    assert maxabs(few[2], t0['obj_feat'].reshape(B, N, C)) < 2e-4
    assert maxabs(few[1], t0['new_mask_preds']) < TOL_LOGIT and maxabs(few[0], t0['cls_score']) < 1e-4
    auto = vkn.ops.stage_forward(*args, **kwd)     # default policy by row tiles: few-row <= 16, launch-per-GEMM 17 .. 21, persistent (two-term fp16 split) from 22 on
    rt = (B * N + 31) // 32
    pick = new if rt >= 22 else (few if rt <= 16 else old)
    assert all(a is None or torch.equal(a, b) for a, b in zip(auto, pick)), 'default policy picks by row count'


def test_few_row_chain_is_tile_shape_and_batch_invariant_bit_for_bit(vkn):
    """The few-row chain picks its tile shape (32 or 64 rows x 32 or 64 columns per workgroup) by row count; a frame must come out the
    same bits whether it runs alone, in a block of 2 (8 row tiles: thin tiles) or among 8 (30 row tiles: fat tiles) — the property the
    sharded clip relies on.  (Round 5: it did not at first — hipcc contracted `bias * count + bias2` into an fma in one instantiation and
    not in another; floating-point contraction is off in csrc/vkn_ksplit.hip since.)  One stage through the C ABI with the raw-gather
    path (count-scaled bias), the video link included."""
    from test_host_logic import _cfg
    C, heads, H, W, N, ff, ncls, B = 256, 8, 8, 16, 117, 2048, 19, 8
    kw = dict(C=C, heads=heads, ffn=ff, ncls=ncls, n_thing=2, n_stuff=17, S=1, up=1, nprop=N - 17)
    case = dict(kw, N=N, H=H, W=W, B=B, seed=4242, video=1)
    head = vkn.build_head(_cfg(True, **kw))
    cfg, sd, x, pf, mp, prev = make_case(case)
    head.load_state_dict(sd, strict=True)
    head = head.to(DEV).eval()
    pack = head.mask_head[0].stage_pack(torch.device(DEV))
    xd, pfd, mpd, pvd = x.to(DEV), pf.reshape(B, N, C).to(DEV), mp.to(DEV), prev.reshape(B, N, C).to(DEV)

    def run(b0, b1):
        dims = head.mask_head[0].make_dims(b1 - b0, N, H, W)
        return vkn.ops.stage_forward(dims, pack, xd[b0:b1], pfd[b0:b1], mpd[b0:b1], prev_obj=pvd[b0:b1], want_track=True, flags=vkn.ops.FLAG_CHAIN_KSPLIT)
    whole = run(0, B)
    for step in (1, 2, 4):
        parts = [run(b, b + step) for b in range(0, B, step)]
        for k, nm in enumerate(('cls', 'masks', 'obj', 'x_feat', 'track')):
            assert torch.equal(torch.cat([p[k] for p in parts], 0), whole[k]), (step, nm)


def test_range_status_word_reports_features_outside_the_f16_split(vkn):
    """|x| >= 65504 (or a non-finite x) turns into inf in the f16 hi/lo split and the MFMA path returns inf / NaN.
    The workspace status word (VKN_STATUS_RANGE, set by the reduction that ends every gather) makes that loud: `check_status()`
    raises VKN_E_RANGE, clears the word, and the next clean call is fine."""
    g, case = load_golden('video_tiny')
    head, (x, pf, mp, prev) = _build_head(vkn, case)
    xd, pfd, mpd, prevd = _cuda(x, pf, mp, prev)
    with torch.no_grad():
        head._head_forward(xd, pfd, mpd, prevd, want_track=True)
        head.check_status()                                  # clean features: nothing raised
        for bad in (7.0e4, float('inf'), float('nan'), -1.0e5):
            xb = xd.clone()
            xb[0, 3, 2, 5] = bad
            head._head_forward(xb, pfd, mpd, prevd, want_track=True)
            with pytest.raises(vkn._lib.VknError) as e:
                head.check_status()
            assert e.value.code == -6, bad
            head.check_status()                              # read-and-clear
        xe = xd.clone()
        xe[0, 3, 2, 5] = 6.5e4                               # the largest magnitudes inside the envelope
        xe[0, 4, 2, 5] = -6.5e4
        head._head_forward(xe, pfd, mpd, prevd, want_track=True)
        head.check_status()
        # a per-stage call reports through the same word
        xb = xd.clone()
        xb[1, 0, 0, 0] = 1.0e6
        head._mask_forward(0, xb, pfd, mpd, [dict()] * case['B'])
        with pytest.raises(vkn._lib.VknError):
            head.check_status()


# ------------------------------------------------------------------------------------------ train-time assignment
@pytest.mark.parametrize('name', ['assign_tiny', 'assign_cfg', 'assign_odd'])
def test_assignment_vs_reference(vkn, name):
    """MaskHungarianAssigner on the GPU: cost matrix vs the reference's (fp32 class accuracy), integer assignment bit-exact."""
    from helpers import load_assign_golden, make_assign_case
    g, case = load_assign_golden(name)
    logits, cls, gt, labels = make_assign_case(case)
    a = vkn.MaskHungarianAssigner(cls_cost=dict(type='FocalLossCost', weight=2.0),
                                  dice_cost=dict(type='DiceCost', weight=4.0, pred_act=True),
                                  mask_cost=dict(type='MaskCost', weight=1.0, pred_act=True))
    dl, dc, dg, dlab = logits.to(DEV), cls.to(DEV), gt.to(DEV), labels.to(DEV)
    cost = a.cost_matrix(dl, dc, dg, dlab)
    assert maxabs(cost, g['cost']) < 2e-5
    res = a.assign(dl, dc, dg, dlab)
    assert res.num_gts == case['G']
    assert np.array_equal(res.gt_inds.cpu().numpy(), g['gt_inds'])          # integer kernel assignments: bit-exact
    assert np.array_equal(res.labels.cpu().numpy(), g['labels'])
    # the margin of the optimum: the second-best assignment is farther than the cost error, so the match is not luck
    from scipy.optimize import linear_sum_assignment
    c64 = g['cost'].astype(np.float64)
    r0, c0 = linear_sum_assignment(c64)
    best = c64[r0, c0].sum()
    got = cost.cpu().numpy().astype(np.float64)
    assert abs(got[r0, c0].sum() - best) < 1e-3


@pytest.mark.parametrize('N,G,ncls,H,W', [(100, 40, 2, 256, 512), (150, 30, 58, 64, 96), (216, 70, 124, 46, 80), (256, 5, 3, 8, 24)],
                         ids=['cfg2', 'n150', 'n216', 'n256'])
def test_assignment_full_resolution(vkn, N, G, ncls, H, W):
    """cfg2 assign resolution (1024x2048 / mask_assign_stride 4 = 256x512, 100 kernels, 40 ground truths) vs fp64 on the device; and
    more than 128 predictions per image (the VIP-Seg sized heads: the two activations take one gather launch each)."""
    lo, cl, gt, lab = (torch.from_numpy(a).to(DEV) for a in synth.assign_inputs(N, G, ncls, H, W, 9))
    a = vkn.MaskHungarianAssigner(cls_cost=dict(type='FocalLossCost', weight=2.0),
                                  dice_cost=dict(type='DiceCost', weight=4.0, pred_act=True),
                                  mask_cost=dict(type='MaskCost', weight=1.0, pred_act=True))
    cost = a.cost_matrix(lo, cl, gt, lab)
    p = lo.double().sigmoid()
    p1, p2, gd = p.clamp(0.001, 1.0).flatten(1), p.clamp(0.01, 1.0).flatten(1), gt.double().flatten(1)
    dice = -(2 * p1 @ gd.t()) / ((p1 * p1).sum(1, keepdim=True) + 1e-3 + gd.sum(1)[None] + 1e-3)
    mcost = -(p2 @ gd.t() + (1 - p2) @ (1 - gd).t()) / (H * W)
    pc = cl.double().sigmoid()
    foc = (-(pc + 1e-12).log() * 0.25 * (1 - pc) ** 2 + (1 - pc + 1e-12).log() * 0.75 * pc ** 2)[:, lab]
    want = 2.0 * foc + 4.0 * dice + mcost
    assert maxabs(cost, want) < 1e-5 * max(2.0, float(want.abs().max()))      # (the focal term is evaluated in fp32)
    res = a.assign(lo, cl, gt, lab)
    from scipy.optimize import linear_sum_assignment
    r0, c0 = linear_sum_assignment(want.cpu().numpy())
    inds = np.zeros(N, dtype=np.int64)
    inds[r0] = c0 + 1
    assert np.array_equal(res.gt_inds.cpu().numpy(), inds)


def test_device_lsap_equals_host_solver_and_scipy(vkn):
    """`vkn_lsap_batch_f32` (one wavefront per matrix, batches of 64 per launch) against the host solver `vkn_lsap_f32` and scipy on 300
    matrices: wide, tall (transposed inside), square, 1 x n, continuous costs, and small-integer costs whose optimum is massively tied
    — the tie rule and scan order decide there, and the results must still be IDENTICAL (integer assignments: bit-exact)."""
    from scipy.optimize import linear_sum_assignment
    rng = np.random.default_rng(5)
    mats = []
    for k in range(300):
        nr, nc = int(rng.integers(1, 129)), int(rng.integers(1, 257))
        if k % 7 == 0:
            nr = nc = int(rng.integers(1, 129))
        if k % 11 == 0:
            nr = 1
        if k % 3 == 0:
            m = rng.integers(0, 4, (nr, nc)).astype(np.float32)                  # ties everywhere
        elif k % 3 == 1:
            m = rng.standard_normal((nr, nc)).astype(np.float32)
        else:
            m = (rng.standard_normal((nr, nc)) * 3).round(1).astype(np.float32)   # some ties
        mats.append(m)
    gts, rows, cols, status = vkn.ops.lsap_device([torch.from_numpy(m).to(DEV) for m in mats])
    assert status.cpu().tolist() == [0] * len(mats)
    for m, g, r, c in zip(mats, gts, rows, cols):
        hr, hc = vkn.ops.lsap(m)
        sr, sc = linear_sum_assignment(m)
        assert np.array_equal(hr, sr) and np.array_equal(hc, sc)
        assert np.array_equal(r.cpu().numpy(), sr) and np.array_equal(c.cpu().numpy(), sc), m.shape
        want = np.zeros(m.shape[0], dtype=np.int64)
        want[sr] = sc + 1
        assert np.array_equal(g.cpu().numpy(), want)
    # status words instead of exceptions: NaN / -inf entries (scipy: ValueError), infeasible (all +inf)
    bad = np.ones((3, 5), np.float32)
    bad[1, 2] = np.nan
    inf = np.full((4, 6), np.inf, np.float32)
    ok = np.arange(12, dtype=np.float32).reshape(3, 4)
    g, r, c, st = vkn.ops.lsap_device([torch.from_numpy(a).to(DEV) for a in (bad, inf, ok)])
    assert st.cpu().tolist() == [1, 2, 0]
    # a failed problem still returns an IN-BOUNDS dummy assignment (row k <-> column 0): the status is read one step late, the
    # indices are used at once
    assert g[0].cpu().tolist() == [1, 1, 1] and r[0].cpu().tolist() == [0, 1, 2] and c[0].cpu().tolist() == [0, 0, 0]
    assert g[1].cpu().tolist() == [1, 1, 1, 1] and r[1].cpu().tolist() == [0, 1, 2, 3] and c[1].cpu().tolist() == [0] * 4
    a = vkn.MaskHungarianAssigner(cls_cost=dict(type='FocalLossCost', weight=2.0), dice_cost=dict(type='DiceCost', weight=4.0, pred_act=True),
                                  mask_cost=dict(type='MaskCost', weight=1.0, pred_act=True))
    a.pending_status.append(st)
    with pytest.raises(ValueError):
        a.check_status()
    a.check_status()   # (drained)


def test_device_and_host_assignment_paths_agree(vkn):
    """The assigner with the device LSAP (default) and with the host LSAP: same AssignResult, and the sampler's index sets built from
    the device tensors equal the ones built from the host copy."""
    N, G, ncls, H, W = 100, 23, 4, 32, 64
    lo, cl, gt, lab = (torch.from_numpy(a).to(DEV) for a in synth.assign_inputs(N, G, ncls, H, W, 4))
    kw = dict(cls_cost=dict(type='FocalLossCost', weight=2.0), dice_cost=dict(type='DiceCost', weight=4.0, pred_act=True),
              mask_cost=dict(type='MaskCost', weight=1.0, pred_act=True))
    ad, ah = vkn.MaskHungarianAssigner(**kw), vkn.MaskHungarianAssigner(**kw)
    ah.lsap = 'host'
    rd, rh = ad.assign(lo, cl, gt, lab), ah.assign(lo, cl, gt, lab)
    assert rd.device_pos_inds is not None and rd.host_pos_inds is None and rh.host_pos_inds is not None
    assert torch.equal(rd.gt_inds, rh.gt_inds) and torch.equal(rd.labels, rh.labels)
    sam = vkn.MaskPseudoSampler()
    sd, sh = sam.sample(rd, lo, gt), sam.sample(rh, lo, gt)
    assert torch.equal(sd.pos_inds, sh.pos_inds) and torch.equal(sd.neg_inds, sh.neg_inds)
    assert torch.equal(sd.pos_assigned_gt_inds, sh.pos_assigned_gt_inds) and torch.equal(sd.pos_gt_labels, sh.pos_gt_labels)
    ad.check_status()


def test_clip_level_assigner_costs_and_layout(vkn):
    """`MaskHungarianAssignerVideo` (knet_vis): costs on the tall [Q, F*H, W] layout WITHOUT the sigmoid clamps of knet's cost classes,
    against the oracle's cost restatement in that flavour; the assignment equals scipy's on the oracle costs."""
    from oracle.knet_oracle import assign_costs
    F, Q, H, W, ncls = 3, 40, 16, 24, 5
    tg = synth.clip_targets(1, F, ncls, H, W, 5, gmin=4, gmax=6)[0]
    logits = torch.from_numpy(synth.normalish((F, Q, H, W), 321, 6.0))          # |z| up to ~12: sigmoid far below both clamps
    cls = torch.from_numpy(synth.normalish((Q, ncls), 322, 2.0))
    a = vkn.MaskHungarianAssignerVideo(cls_cost=dict(type='FocalLossCost', weight=2.0),
                                       dice_cost=dict(type='DiceCost', weight=4.0, pred_act=True),
                                       mask_cost=dict(type='MaskCost', weight=1.0, pred_act=True))
    masks = [torch.from_numpy(m).to(DEV) for m in tg['gt_masks']]
    lab, ids = torch.from_numpy(tg['gt_labels']).to(DEV), torch.from_numpy(tg['gt_instance_ids']).to(DEV)
    res, gt_tall = a.assign(logits.to(DEV), cls.to(DEV), masks, lab, ids)
    clip, labels, inst = a.clip_instances(F, [m.cpu() for m in masks], lab.cpu(), ids.cpu())
    assert torch.equal(gt_tall.cpu(), clip.reshape(len(inst), F * H, W))
    tall = a.tall(logits)
    want = assign_costs(tall, cls, clip.reshape(len(inst), F * H, W), labels, dice_pred_min=0.0, mask_pred_min=0.0)
    got = a.cost_matrix(tall.to(DEV), cls.to(DEV), gt_tall, labels.to(DEV))
    assert maxabs(got, want) < 2e-5
    clamped = assign_costs(tall, cls, clip.reshape(len(inst), F * H, W), labels)
    assert maxabs(clamped, want) > 1e-4                                         # the two flavours really differ on these inputs
    from scipy.optimize import linear_sum_assignment
    r0, c0 = linear_sum_assignment(want.numpy())
    inds = np.zeros(Q, dtype=np.int64)
    inds[r0] = c0 + 1
    assert np.array_equal(res.gt_inds.cpu().numpy(), inds)
    assert np.array_equal(res.labels.cpu().numpy()[r0], labels.numpy()[c0])


@pytest.mark.parametrize('name', ['pan_tiny', 'pan_cfg'])
def test_segment_boxes_for_tracking(vkn, name):
    """bbox output of the panoptic pipeline == tensor_mask2box(panoptic_seg == id) on the reference's panoptic map for every
    thing segment (the tracker's input), integer-exact wherever the GPU map equals the reference map."""
    from helpers import PAN_CFG, load_pan_golden, make_pan_case, run_pan_oracle
    g, case = load_pan_golden(name)
    cls, logits, meta = make_pan_case(case)
    seg, info, nseg, bbox = vkn.ops.panoptic_joint(cls.to(DEV), logits.to(DEV), case['Np'], case['T'], case['Np'],
                                                   PAN_CFG['instance_score_thr'], PAN_CFG['overlap_thr'], meta['img_shape'][:2],
                                                   meta['batch_input_shape'], meta['ori_shape'][:2], upsample_stride=case['up'],
                                                   want_bbox=True)
    torch.cuda.synchronize()
    seg, info, bbox = seg.cpu().numpy(), info.cpu().numpy(), bbox.cpu().numpy()
    head_like = type('H', (), dict(num_thing_classes=case['T']))()
    from importlib import import_module
    KIH = import_module('video_k_net_amd.kernel_iter_head').KernelIterHead
    for b in range(case['B']):
        r = run_pan_oracle(case, b)
        # boxes of the GPU's own map, through the oracle's restatement of get_things_id_for_tracking + tensor_mask2box
        infos = KIH._segments_info(head_like, info[b])
        want = O.things_for_tracking(seg[b], infos)
        got = KIH.things_for_tracking(head_like, info[b], bbox[b])
        assert got[0] == want[0] and got[1] == want[1] and len(got[0]) > 0
        assert np.array_equal(got[2], np.asarray(want[2], dtype=np.float32))
        assert np.allclose(got[3], want[3], rtol=0, atol=0)
        if np.array_equal(seg[b], g['panoptic_seg'][b]):     # identical maps -> identical boxes as the reference's map gives
            ref = O.things_for_tracking(g['panoptic_seg'][b], r['segments_info'])
            assert np.array_equal(got[2], np.asarray(ref[2], dtype=np.float32))
        rej = info[b, :, 2] == 0
        assert (bbox[b][rej] == np.array([-1, -1, 10, 10])).all()


def test_batch_invariance(vkn):
    """Frames are independent: a 19-frame call gives, bit for bit, what single-frame calls give (and the x`up` output is the
    stand-alone upsample of the final logits)."""
    _, case = load_golden('video_tiny')
    head, (x, pf, mp, prev) = _build_head(vkn, case)
    B, N, C, H, W = 19, case['N'], case['C'], case['H'], case['W']
    xs = _rand((B, C, H, W), 901).to(DEV)
    pfs = _rand((B, N, C), 902).to(DEV)
    mps = _rand((B, N, H, W), 903, 4.0).to(DEV)
    dims = head.mask_head[0].make_dims(B, N, H, W)
    packs = [h.stage_pack(torch.device(DEV)) for h in head.mask_head]
    obj, cls, masks, scaled, _ = vkn.ops.head_forward(dims, packs, xs, pfs, mps, None, case['up'])
    assert torch.equal(scaled, vkn.ops.upsample_bilinear(masks, case['up']))
    d1 = head.mask_head[0].make_dims(1, N, H, W)
    for b in (0, 7, 8, 18):
        o1, c1, m1, s1, _ = vkn.ops.head_forward(d1, packs, xs[b:b + 1], pfs[b:b + 1], mps[b:b + 1], None, case['up'])
        assert torch.equal(m1[0], masks[b]) and torch.equal(s1[0], scaled[b]) and torch.equal(c1[0], cls[b])


def test_panoptic_selection_many_thing_classes(vkn):
    """VIP-Seg-like class layout (58 thing classes, 100 proposals -> 5800 (proposal, class) candidates, 66 stuff kernels): the
    top-k / sort / merge-order kernels against the oracle's torch.topk / sort / argsort, bit-exact."""
    N, Np, T, ncls, Hm, Wm = 166, 100, 58, 124, 16, 24
    cls_np, logit_np = synth.panoptic_inputs(1, N, Np, ncls, Hm, Wm, 321)
    meta = dict(img_shape=(64, 96, 3), batch_input_shape=(64, 96), ori_shape=(64, 96, 3))
    seg, info, nseg = vkn.ops.panoptic_joint(torch.from_numpy(cls_np).to(DEV), torch.from_numpy(logit_np).to(DEV), Np, T, 100, 0.25,
                                             0.6, (64, 96), (64, 96), (64, 96), upsample_stride=4)
    with torch.no_grad():
        r = O.panoptic_joint(torch.from_numpy(cls_np)[0], torch.from_numpy(logit_np)[0], Np, T, 100, 0.25, 0.6, meta,
                             upsample_stride=4)
    info = info[0].cpu().numpy()
    assert np.array_equal(info[:, 0], r['rows'].numpy()) and np.array_equal(info[:, 1], r['total_labels'].numpy())
    assert np.array_equal(info[:, 5].view(np.float32), r['total_scores'].numpy())
    assert np.array_equal(info[:, 2], r['seg_of'].numpy()) and int(nseg[0]) == len(r['segments_info'])
    near = r['margin'].numpy() < 1e-6
    assert not ((seg[0].cpu().numpy() != r['panoptic_seg'].numpy()) & ~near).any()


def test_clip_link_in_one_call(vkn):
    """VKN_FLAG_CLIP_LINK: frame b links to frame b - 1 of the same call, frame 0 to the given previous kernels — identical to the
    two-step path (head, then track_link on the shifted kernels)."""
    _, case = load_golden('video_cfg')
    head, (x, pf, mp, prev) = _build_head(vkn, case)
    T, N, C = 5, case['N'], case['C']
    xs = _rand((T, C, case['H'], case['W']), 931).to(DEV)
    pfs = _rand((T, N, C), 932).to(DEV)
    mps = _rand((T, N, case['H'], case['W']), 933, 4.0).to(DEV)
    first = _rand((1, N, C), 934).to(DEV)
    dims = head.mask_head[0].make_dims(T, N, case['H'], case['W'])
    packs = [h.stage_pack(torch.device(DEV)) for h in head.mask_head]
    a = vkn.ops.head_forward(dims, packs, xs, pfs, mps, None, case['up'], clip_first_prev=first)
    b = vkn.ops.head_forward(dims, packs, xs, pfs, mps, None, case['up'])
    prevs = torch.cat([first, b[0][:-1]], 0)
    track = vkn.ops.track_link(dims, packs[-1], b[0], prevs)
    for u, v in zip(a[:4], b[:4]):
        assert torch.equal(u, v)
    assert torch.equal(a[4], track)
    o = head.clip_forward(xs, pfs.reshape(T, N, C, 1, 1), mps, first_previous_obj_feats=first.reshape(1, N, C, 1, 1))
    assert torch.equal(o[4].reshape(T, N, C), track) and torch.equal(o[2], b[2])


def test_side_stream_link_equals_serial_link(vkn):
    """The fused head runs the tracking link on the library's side stream (forked where the last stage's kernels are final,
    joined before the call returns); VKN_FLAG_SERIAL_LINK keeps it on the caller's stream.  Same kernels on the same operands
    -> every output bit-identical, in clip mode and with a per-frame previous tensor, repeated to catch ordering races."""
    _, case = load_golden('video_cfg')
    head, (x, pf, mp, prev) = _build_head(vkn, case)
    T, N, C = 6, case['N'], case['C']
    xs = _rand((T, C, case['H'], case['W']), 941).to(DEV)
    pfs = _rand((T, N, C), 942).to(DEV)
    mps = _rand((T, N, case['H'], case['W']), 943, 4.0).to(DEV)
    first = _rand((1, N, C), 944).to(DEV)
    prevs = _rand((T, N, C), 945).to(DEV)
    dims = head.mask_head[0].make_dims(T, N, case['H'], case['W'])
    packs = [h.stage_pack(torch.device(DEV)) for h in head.mask_head]
    for kw in ({'clip_first_prev': first}, {'prev_obj': prevs}):
        ref = vkn.ops.head_forward(dims, packs, xs, pfs, mps, kw.get('prev_obj'), case['up'],
                                   clip_first_prev=kw.get('clip_first_prev'), want_track=True, flags=vkn.ops.FLAG_SERIAL_LINK)
        for _ in range(4):
            out = vkn.ops.head_forward(dims, packs, xs, pfs, mps, kw.get('prev_obj'), case['up'],
                                       clip_first_prev=kw.get('clip_first_prev'), want_track=True)
            assert out[4] is not None
            for u, v in zip(out, ref):
                assert (u is None and v is None) or torch.equal(u, v)


@pytest.mark.parametrize('chain', ['ksplit', 'launches', 'policy'])
def test_block_step_with_neighbour_link_equals_whole_clip(vkn, chain):
    """bench.py --gpus N: every rank runs its contiguous block of the clip as ONE call with the in-call clip link and re-links only
    its frame 0 to the previous rank's last kernels (dist.neighbour_last_kernels).  Emulated on one GPU with two blocks: every
    output of the two block steps equals the whole-clip call — BIT FOR BIT when both run the same form of the [N x C] chain (every
    form is row-independent with a fixed summation order: batch-invariant; the one-frame re-link is pinned to the call's form through
    vkn_track_link_flags_f32), to fp32 rounding when the row-count policy gives the blocks (11 row tiles: few-row chain)
    another form than the whole clip (22 row tiles: one launch per GEMM)."""
    _, case = load_golden('video_cfg')
    head, _ = _build_head(vkn, case)
    T, N, C, H, W = 6, case['N'], case['C'], case['H'], case['W']
    xs = _rand((T, C, H, W), 951).to(DEV)
    pfs = _rand((T, N, C), 952).to(DEV)
    mps = _rand((T, N, H, W), 953, 4.0).to(DEV)
    first = _rand((1, N, C), 954).to(DEV)
    packs = [h.stage_pack(torch.device(DEV)) for h in head.mask_head]
    mk = head.mask_head[0].make_dims
    fl = dict(ksplit=vkn.ops.FLAG_CHAIN_KSPLIT, launches=vkn.ops.FLAG_CHAIN_LAUNCHES, policy=0)[chain]
    whole = vkn.ops.head_forward(mk(T, N, H, W), packs, xs, pfs, mps, None, case['up'], clip_first_prev=first, flags=fl)
    h = T // 2
    blocks = []
    prev_last = None
    for r, (b0, b1) in enumerate(((0, h), (h, T))):
        out = vkn.ops.head_forward(mk(b1 - b0, N, H, W), packs, xs[b0:b1], pfs[b0:b1], mps[b0:b1], None, case['up'],
                                   clip_first_prev=first, flags=fl)
        if r > 0:   # what rank r does after the neighbour hand-over
            out[4][0:1].copy_(vkn.ops.track_link(mk(1, N, H, W), packs[-1], out[0][0:1], prev_last, flags=fl))   # pinned to the call's form
        prev_last = out[0][-1:].clone()
        blocks.append(out)
    for k in range(5):
        got = torch.cat([b[k] for b in blocks], 0)
        if chain == 'policy' and (T * N + 31) // 32 > 16 >= (h * N + 31) // 32:
            # two forms of the chain: the teacher-forced distance (2e-5 relative per stage) grows through three free-running stages only
            # where a near-threshold mask bit flips; this small case (16x32 features) has none
            assert maxabs(got, whole[k]) < 2e-4 * max(1.0, float(whole[k].abs().max())), k
        else:
            assert torch.equal(got, whole[k]), k


def test_softmax_classification_head_runs_stage_by_stage_vs_oracle(vkn):
    """`loss_cls.use_sigmoid=False` (reference knet/det/kernel_iter_head.py:309-310: `cls_score.softmax(-1)[..., :-1]`, `fc_cls` with
    one more output; no shipped config): the reference entry point works — stage by stage, because the fused call applies the sigmoid
    in its last epilogue — and matches the oracle; the fused extension API refuses loudly."""
    import dataclasses
    from helpers import cfg_of
    from oracle import synth
    from oracle.knet_oracle import head_param_shapes, iter_head_mask_preds
    from test_host_logic import _cfg
    _, case = load_golden('det_tiny')
    ocfg = dataclasses.replace(cfg_of(case), use_sigmoid_cls=False)
    sd = {k: torch.from_numpy(v) for k, v in synth.state_dict_like(head_param_shapes(ocfg), case['seed']).items()}
    x, pf, mp = (torch.from_numpy(a) for a in synth.head_inputs(case['B'], case['N'], case['C'], case['H'], case['W'], case['seed']))
    with torch.no_grad():
        r_obj, r_cls, r_masks, _, _ = iter_head_mask_preds(sd, x, pf, mp, ocfg)
    head = vkn.build_head(_cfg(False, C=case['C'], heads=case['heads'], ffn=case['ffn'], ncls=case['ncls'], n_thing=case['n_thing'],
                               n_stuff=case['n_stuff'], S=case['S'], up=case['up'], nprop=case['nprop'],
                               mask_over=dict(loss_cls=dict(type='CrossEntropyLoss', use_sigmoid=False, loss_weight=1.0))))
    head.load_state_dict(sd, strict=True)
    head = head.to(DEV).eval()
    xd, pfd, mpd = _cuda(x, pf, mp)
    with torch.no_grad():
        obj, cls, masks, scaled = head.simple_test_mask_preds(xd, pfd, mpd, None, [dict()] * case['B'])
    assert tuple(cls.shape) == (case['B'], case['N'], case['ncls'])
    assert maxabs(cls, r_cls) < 1e-5
    assert maxabs(obj, r_obj) < 1e-4
    assert maxabs(masks, r_masks) < TOL_LOGIT
    with pytest.raises(NotImplementedError):
        head._head_forward(xd, pfd, mpd)


def test_masks_at_another_resolution_are_resized_like_the_reference(vkn):
    """Reference knet/det/kernel_update_head.py:182-188 (incoming masks at another resolution than x are resized bilinearly before the
    gather) and :268-273 (`mask_shape` resizes the new logits): dead in shipped configs, built since round 4.  Fused call and
    stage-by-stage path against the oracle on half-resolution incoming masks."""
    import torch.nn.functional as F
    from helpers import cfg_of, make_case
    from oracle.knet_oracle import iter_head_mask_preds
    _, case = load_golden('det_tiny')
    head, (x, pf, mp, _) = _build_head(vkn, case)
    cfg, sd, *_ = make_case(case)
    mp_small = F.avg_pool2d(mp, 2) * 3.0                      # a different tensor at half the resolution
    with torch.no_grad():
        r_obj, r_cls, r_masks, _, _ = iter_head_mask_preds(sd, x, pf, mp_small, cfg_of(case))
    xd, pfd, mpd = _cuda(x, pf, mp_small)
    with torch.no_grad():
        obj, cls, masks, scaled = head.simple_test_mask_preds(xd, pfd, mpd, None, [dict()] * case['B'])
        r = head._mask_forward(0, xd, pfd, mpd, [dict()] * case['B'])
        cls1, m1, obj1 = head.mask_head[0](xd, pfd, mpd, mask_shape=(2 * case['H'], 2 * case['W']))
    assert maxabs(obj, r_obj) < 1e-4 and maxabs(cls, r_cls) < 1e-5 and maxabs(masks, r_masks) < TOL_LOGIT
    assert tuple(m1.shape[-2:]) == (2 * case['H'], 2 * case['W'])
    assert maxabs(m1, F.interpolate(r['mask_preds'], scale_factor=2, mode='bilinear', align_corners=False)) < 1e-5


def test_status_word_survives_the_ops_that_share_the_workspace(vkn):
    """The stage-shaped entry points keep a status word in the first 256 bytes of their workspace; gather / decode / kernel-init /
    assignment calls use THEIR workspace from offset 0.  The binding shares one buffer per stream between both kinds — it has to hand
    the second kind the part behind the header (found by the soak in round 4: `check_status` raised after a kernel-init pass)."""
    _, case = load_golden('det_tiny')
    head, (x, pf, mp, _) = _build_head(vkn, case)
    xd, pfd, mpd = _cuda(x, pf, mp)
    B, N, C, H, W = case['B'], case['N'], case['C'], case['H'], case['W']
    with torch.no_grad():
        head.simple_test_mask_preds(xd, pfd, mpd, None, [dict()] * B)
        head.check_status()
        for _ in range(3):      # scratch-heavy ops between two head calls, on the same stream
            vkn.ops.kernel_init(xd, xd * 0.5, torch.randn(N, C, 1, 1, device=DEV), torch.randn(5, C, 1, 1, device=DEV), torch.randn(5, device=DEV))
            vkn.ops.mask_decode(xd, torch.randn(B, N, C, device=DEV))
            vkn.ops.mask_gather(xd, mpd, 0.5)
            vkn.ops.mask_gather_real(xd, torch.rand(B, N, H, W, device=DEV))
            vkn.ops.linear(torch.randn(64, 256, device=DEV), torch.randn(32, 256, device=DEV), ksplit=4)
        head.check_status()                                  # must not see their scratch data
        head.simple_test_mask_preds(xd, pfd, mpd, None, [dict()] * B)
        head.check_status()
        bad = xd.clone()
        bad[0, 0, 0, 0] = float('inf')
        head.simple_test_mask_preds(bad, pfd, mpd, None, [dict()] * B)
        with pytest.raises(vkn._lib.VknError):
            head.check_status()                              # ... and still reports a real one
        head.check_status()                                  # read-and-clear

"""GPU (-m gpu): the "update" video heads — `previous_link="update_dynamic_cov"` (+ `previous_type="update"` | "ffn"), shipped in
configs/det/video_knet_kitti_step/video_knet_s3_swin{b,l}_*_joint_update.py:98-100 and ..._update_conv_short_track_fc.py:95-97 —
against goldens captured from the reference's own `VideoKernelUpdateHead` (oracle/gen_golden.py: video_upd_*), against the oracle,
and clip-batched inference against the reference's frame-by-frame walk of a video.

With previous_link the LAST stage's incoming kernels of frame t are rewritten from frame t-1's final kernels
(knet/video/kernel_update_head.py:324-348), i.e. masks depend on the previous frame: `clip_forward` keeps every pass over x batched
and runs only the last stage's [N x C] chain frame by frame (vkn_head_forward_link_f32).
"""
import numpy as np
import pytest
import torch

from helpers import cfg_of, load_golden, make_case, maxabs, run_oracle
from oracle import knet_oracle as O
from oracle import synth
from test_gpu_parity import _build_head, _cuda

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
CASES = ['video_upd_tiny', 'video_updffn_tiny', 'video_upd_cfg', 'video_latt_upd_tiny', 'video_updobj_tiny', 'video_latt_updobj_tiny']


def _rand(shape, salt, std=1.0):
    return torch.from_numpy(synth.normalish(shape, salt, std))


@pytest.mark.parametrize('name', CASES)
def test_state_dict_and_modules_match_reference(vkn, name):
    g, case = load_golden(name)
    head, _ = _build_head(vkn, case)
    sd = head.state_dict()
    assert sorted(sd) == list(g['keys'])
    assert [str(tuple(sd[k].shape)) for k in sorted(sd)] == list(g['shapes'])


@pytest.mark.parametrize('flags', [0, 3], ids=['mfma', 'allexact'])
@pytest.mark.parametrize('name', CASES)
def test_update_head_vs_reference_golden(vkn, name, flags):
    """Fused S-stage call, every frame with its own (synthetic) previous kernels: kernels, cls, masks and the tracking embedding
    against the REFERENCE's outputs and the oracle."""
    g, case = load_golden(name)
    head, (x, pf, mp, prev) = _build_head(vkn, case)
    xd, pfd, mpd, prevd = _cuda(x, pf, mp, prev)
    with torch.no_grad():
        obj, cls, masks, scaled, track = head._head_forward(xd, pfd, mpd, prevd, want_track=True, flags=flags)
    assert maxabs(obj, g['object_feats']) < 1e-4
    assert maxabs(cls, g['cls_score']) < 1e-5
    assert maxabs(masks, g['mask_preds']) < 1e-3
    margin = np.abs(g['mask_preds']) > 2e-3
    assert np.array_equal((masks.cpu().numpy() > 0)[margin], (g['mask_preds'] > 0)[margin]), 'binary masks must be bit-exact'
    assert maxabs(track, g['track']) < 1e-4
    o_obj, o_cls, o_masks, o_scaled, o_track = run_oracle(case)
    assert maxabs(obj, o_obj) < 1e-4 and maxabs(masks, o_masks) < 1e-3 and maxabs(track, o_track) < 1e-4
    assert maxabs(scaled, o_scaled) < 1e-3
    # the public API (track dropped, reference :529-564) and the serial-link variant return the same bits
    with torch.no_grad():
        out = head.simple_test_mask_preds_plus_previous(xd, pfd, mpd, None, [dict()] * case['B'], previous_obj_feats=prevd,
                                                        return_track=True)
        ser = head._head_forward(xd, pfd, mpd, prevd, want_track=True, flags=flags | vkn.ops.FLAG_SERIAL_LINK)
    if flags == 0:
        assert torch.equal(out[2], masks) and torch.equal(out[4], track)
    for u, v in zip(ser, (obj, cls, masks, scaled, track)):
        assert torch.equal(u, v)


@pytest.mark.parametrize('name', CASES)
def test_update_stage_module_api_vs_reference(vkn, name):
    """`_mask_forward` stage by stage (the reference's per-stage API): the last stage gets previous_obj_feats."""
    g, case = load_golden(name)
    head, (x, pf, mp, prev) = _build_head(vkn, case)
    xd, obj, masks, prevd = _cuda(x, pf, mp, prev)
    with torch.no_grad():
        for s in range(case['S']):
            last = s == case['S'] - 1
            r = head._mask_forward(s, xd, obj, masks, [dict()] * case['B'], previous_obj_feats=prevd if last else None)
            obj, masks = r['object_feats'], r['mask_preds']
            if not last:   # the golden's per-stage intermediates are the no-previous walk: identical up to the last stage
                assert maxabs(r['cls_score'], g[f's{s}_cls']) < 1e-4 and maxabs(obj, g[f's{s}_obj']) < 1e-4
    assert maxabs(obj, g['object_feats']) < 1e-4 and maxabs(masks, g['mask_preds']) < 1e-3
    assert maxabs(r['object_feats_track'], g['track']) < 1e-4
    assert maxabs(r['cls_score'].sigmoid(), g['cls_score']) < 1e-5


@pytest.mark.parametrize('name', CASES)
def test_clip_forward_vs_reference_video_walk(vkn, name):
    """The B frames as consecutive frames of ONE video.  Reference: frame by frame, frame t's last stage linked to frame t-1's final
    kernels (knet/video/knet_quansi_dense_embed_fc_joint_train.py:505-525; golden `clip_*`).  Here: one `clip_forward` call —
    and it must also equal OUR frame-by-frame walk bit for bit (same kernels on the same operands)."""
    g, case = load_golden(name)
    head, (x, pf, mp, _) = _build_head(vkn, case)
    xd, pfd, mpd = _cuda(x, pf, mp)
    with torch.no_grad():
        obj, cls, masks, scaled, track = head.clip_forward(xd, pfd, mpd)
        memo, walk = None, []
        for t in range(case['B']):
            o = head.simple_test_mask_preds_plus_previous(xd[t:t + 1], pfd[t:t + 1], mpd[t:t + 1], None, [dict()],
                                                          previous_obj_feats=memo, return_track=True)
            memo = o[0]
            walk.append(o)
    for t in range(case['B']):
        assert maxabs(obj[t:t + 1], g[f'clip_obj{t}']) < 2e-4, t
        assert maxabs(cls[t:t + 1], g[f'clip_cls{t}']) < 1e-5, t
        if f'clip_mask{t}' in g:
            assert maxabs(masks[t:t + 1], g[f'clip_mask{t}']) < 1e-3, t
        else:
            rs = masks[t:t + 1].double().sum(dim=(-1, -2)).cpu().numpy()
            assert np.max(np.abs(rs - g[f'clip_mask_rowsum{t}'])) < 1e-3 * case['H'] * case['W'] / 16, t
        if f'clip_track{t}' in g:
            assert maxabs(track[t:t + 1], g[f'clip_track{t}']) < 2e-4, t
        else:
            assert torch.equal(track[t], obj[t])       # first frame: the kernels themselves are the tracking feature (:474-475)
        w = walk[t]
        assert torch.equal(obj[t:t + 1], w[0]) and torch.equal(cls[t:t + 1], w[1]) and torch.equal(masks[t:t + 1], w[2])
        assert torch.equal(scaled[t:t + 1], w[3])
        if w[4] is not None:
            assert torch.equal(track[t:t + 1], w[4])


def test_clip_forward_handoffs_agree(vkn):
    """Frame-sequential last stage behind each of the three stage hand-offs (fused pass / bit words / fp32 logits) and with the
    exact-fp32 kernels: the hand-offs are bit-identical to each other."""
    _, case = load_golden('video_upd_cfg')
    head, _ = _build_head(vkn, case)
    T, N, C, H, W = 4, case['N'], case['C'], case['H'], case['W']
    xs, pfs = _rand((T, C, H, W), 961).to(DEV), _rand((T, N, C), 962).to(DEV)
    mps, first = _rand((T, N, H, W), 963, 4.0).to(DEV), _rand((1, N, C), 964).to(DEV)
    dims = head.mask_head[0].make_dims(T, N, H, W)
    packs = [h.stage_pack(torch.device(DEV)) for h in head.mask_head]
    pre, trk, src = head.mask_head[-1].link_packs(torch.device(DEV))
    assert pre is not None and trk is not None and src == 1
    outs = [vkn.ops.head_forward(dims, packs, xs, pfs, mps, None, case['up'], clip_first_prev=first, link_pre=pre, link_track=trk,
                                 track_src=src, flags=fl) for fl in (0, vkn.ops.FLAG_BITS_HANDOFF, vkn.ops.FLAG_LOGITS_HANDOFF)]
    for o in outs[1:]:
        for u, v in zip(o, outs[0]):
            assert torch.equal(u, v)
    # the link really changes the masks of frames >= 1 (otherwise this test would pass with the link ignored)
    plain = vkn.ops.head_forward(dims, packs, xs, pfs, mps, None, case['up'])
    assert torch.equal(plain[2][0], plain[2][0]) and not torch.equal(plain[2][1:], outs[0][2][1:])


@pytest.mark.parametrize('with_updator', [False, True])
def test_link_block_vs_oracle(vkn, with_updator):
    """`vkn_link_block_f32` alone against the oracle's restatement of the block."""
    _, case = load_golden('video_upd_cfg')
    cfg, sd, *_ = make_case(case)
    head, _ = _build_head(vkn, case)
    last = head.mask_head[-1]
    B, N, C = 3, case['N'], case['C']
    cur, prev, uf = _rand((B, N, C), 971), _rand((B, N, C), 972), _rand((B, N, C), 973, 30.0)
    named = dict(last.named_parameters())
    if with_updator:
        pack = vkn.ops.link_pack(named, torch.device(DEV), *last._link_names('link'))
    else:
        pack = vkn.ops.link_pack(named, torch.device(DEV), None, 'attention_previous_link', 'attention_previous_norm_link',
                                 'link_ffn_link', 'link_ffn_norm_link')
    dims = last.make_dims(B, N, case['H'], case['W'])
    out = vkn.ops.link_block(dims, pack, cur.to(DEV), prev.to(DEV), uf.to(DEV) if with_updator else None)
    pfx = f'mask_head.{case["S"] - 1}'
    ref = O.link_block(sd, pfx, '_link', with_updator, uf, cur, prev, cfg)
    assert maxabs(out, ref) < 1e-4


@pytest.mark.parametrize('ptype', ['ffn', 'update', 'update_obj'])
@pytest.mark.parametrize('plink', [None, 'update_dynamic_cov', 'link_atten'])
def test_every_link_combination_clip_equals_frame_by_frame(vkn, plink, ptype):
    """All nine (previous_link, previous_type) combinations the reference's ctor accepts — only three are shipped configs and have
    goldens — through the in-call clip (frame-sequential last stage, batched tracking link) and frame by frame: every output
    bit-identical.  (A soak run found the unshipped `link_atten` + `update` pair reading an x_feat the clip path had not written.)"""
    torch.manual_seed(3)
    for C in (64, 128):
        head = vkn.build_head(vkn.configs.roi_head_cfg(True, C=C, heads=8, ffn=128, ncls=5, n_thing=2, n_stuff=3, S=2, up=2, nprop=20,
                                                       mask_over=dict(previous_link=plink, previous_type=ptype)))
        head.init_weights()
        head = head.to(DEV).eval()
        for T, H, W in ((2, 8, 8), (3, 16, 16), (4, 8, 16)):
            N = 23
            x, pf = torch.randn(T, C, H, W, device=DEV), torch.randn(T, N, C, 1, 1, device=DEV)
            mp, first = torch.randn(T, N, H, W, device=DEV) * 3, torch.randn(1, N, C, 1, 1, device=DEV)
            with torch.no_grad():
                clip = head.clip_forward(x, pf, mp, first)
                prev = first
                for t in range(T):
                    one = head.clip_forward(x[t:t + 1], pf[t:t + 1], mp[t:t + 1], prev)
                    for name, u, v in zip(('obj', 'cls', 'masks', 'scaled', 'track'), one, clip):
                        assert torch.equal(u[0], v[t]), (plink, ptype, C, T, t, name)
                    prev = one[0][0:1]


@pytest.mark.parametrize('name', ['video_upd_tiny', 'video_updffn_tiny', 'video_upd_cfg', 'video_latt_upd_tiny'])
def test_clip_in_phases_equals_the_whole_call_and_shards_over_blocks(vkn, name):
    """`vkn_head_forward_link_f32` in phases (VKN_FLAG_PHASE_A / B / C) — what a rank of a frame-sharded clip runs
    around its ONE receive and ONE send (dist.linked_block_forward).  (1) A, B, C through `linked_block_phases` equal the unphased
    clip call bit for bit; (2) the clip cut into two blocks, the second one's phase B started from the first one's last kernels —
    two ranks emulated on one GPU — equals the whole clip bit for bit."""
    from importlib import import_module
    d = import_module('video_k_net_amd.dist')
    g, case = load_golden(name)
    head, _ = _build_head(vkn, case)
    T, N, C, H, W = 5, case['N'], case['C'], case['H'], case['W']
    xs = _rand((T, C, H, W), 1951).to(DEV)
    pfs = _rand((T, N, C, 1, 1), 1952).to(DEV)
    mps = _rand((T, N, H, W), 1953, 4.0).to(DEV)
    first = _rand((1, N, C), 1954).to(DEV)
    # bit-for-bit equality of blocks and whole clip holds per FORM of the [N x C] chain (each is row-independent with a fixed summation
    # order); the row-count policy would give the 5-frame clip (19 row tiles at N = 117) another form than its 2- / 3-frame blocks, so
    # the config-width case pins one form (tests/test_gpu_parity.py::test_block_step_with_neighbour_link_equals_whole_clip has the
    # cross-form tolerance case)
    if case['C'] == 256:
        for st in head.mask_head:
            st.vkn_flags = vkn.ops.FLAG_CHAIN_KSPLIT
    with torch.no_grad():
        whole = head.clip_forward(xs, pfs, mps, first_previous_obj_feats=first.reshape(1, N, C, 1, 1))
        phased = d.linked_block_forward(head.linked_block_phases(xs, pfs, mps), first)        # no process group: one rank, three calls
        for k in range(5):
            assert torch.equal(phased[k].reshape(whole[k].shape), whole[k]), k
        # two blocks = two ranks: block 1's phase A may run before block 0 is done (separate output tensors; the workspace is
        # per (thread, stream), so the emulation runs the blocks one after the other)
        h = 2
        run0 = head.linked_block_phases(xs[:h], pfs[:h], mps[:h])
        run0('A', None)
        k0 = run0('B', first)
        out0 = run0('C', first)
        run1 = head.linked_block_phases(xs[h:], pfs[h:], mps[h:])
        hand_over = k0[-1:].clone()                                   # what rank 0 sends to rank 1
        run1('A', None)
        run1('B', hand_over)
        out1 = run1('C', hand_over)
        for k in range(5):
            got = torch.cat([out0[k], out1[k]], 0)
            assert torch.equal(got.reshape(whole[k].shape), whole[k]), k

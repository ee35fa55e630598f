#!/usr/bin/env python3
"""GPU soak test: random shapes / seeds through the C ABI against the CPU oracle (test infrastructure), beyond the fixed cases of
tests/.  usage: soak.py [seconds per section]   — prints one line per section, exits non-zero on the first mismatch."""
import os
import sys
import time

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import vkn_import  # noqa: E402
from oracle import knet_oracle as O  # noqa: E402
from oracle import synth  # noqa: E402

vkn = vkn_import.load()
dev = 'cuda:0'
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 20.0
rng = np.random.default_rng(int(os.environ.get('VKN_SOAK_SEED', '12345')))   # (tools only: another seed = other shapes)


LAST = {}   # the case a section is working on (printed when it dies)


def section(name, fn):
    t0, n = time.time(), 0
    while time.time() - t0 < budget:
        try:
            fn()
        except BaseException:
            print(f'{name}: FAILED in trial {n}, case {LAST.get("tag")}', flush=True)
            raise
        n += 1
    print(f'{name:28s} {n:5d} random trials OK')


def t_gather_decode():
    B, N = int(rng.integers(1, 4)), int(rng.integers(1, 200))
    C = int(rng.choice([32, 64, 128, 256]))
    H, W = int(rng.integers(1, 40)), int(rng.integers(1, 40))
    if H * W < 2:
        W = 2
    x = torch.randn(B, C, H, W, device=dev)
    m = torch.randn(B, N, H, W, device=dev) * 3
    k = torch.randn(B, N, C, device=dev)
    xr, cnt = vkn.ops.mask_gather(x, m)
    bits = (m >= vkn.ops.thr_logit(0.5)).double()
    want = torch.einsum('bnhw,bchw->bnc', bits, x.double())
    assert float((xr.double() - want).abs().max()) < 1e-4 * max(1.0, float(want.abs().max())), ('gather', B, N, C, H, W)
    assert torch.equal(cnt.double(), bits.flatten(2).sum(2)), ('count', B, N, C, H, W)
    out = vkn.ops.mask_decode(x, k)
    wd = torch.einsum('bnc,bchw->bnhw', k.double(), x.double())
    assert float((out.double() - wd).abs().max()) < 2e-4 * max(1.0, float(wd.abs().max())), ('decode', B, N, C, H, W)


def t_upsample():
    S = int(rng.choice([2, 3, 4]))
    planes, H, W = int(rng.integers(1, 20)), int(rng.integers(1, 50)), int(rng.integers(1, 70))
    a = torch.randn(1, planes, H, W, device=dev)
    got = vkn.ops.upsample_bilinear(a, S)
    want = F.interpolate(a, scale_factor=S, mode='bilinear', align_corners=False)
    assert float((got - want).abs().max()) < 1e-5, ('upsample', S, planes, H, W)


def t_panoptic():
    N, Np, T = 30, 20, int(rng.integers(1, 4))
    ncls = T + (N - Np)
    Hm, Wm, up = int(rng.integers(4, 20)), int(rng.integers(4, 30)), int(rng.choice([1, 2, 4]))
    Ha, Wa = Hm * up, Wm * up
    f = float(rng.choice([1.0, 2.0, 1.5]))
    Hb, Wb = int(Ha * f), int(Wa * f)
    h, w = int(rng.integers(max(1, Hb - 5), Hb + 1)), int(rng.integers(max(1, Wb - 5), Wb + 1))
    mode = int(rng.integers(0, 3))
    Ho, Wo = (h, w) if mode == 0 else ((int(h * 1.5), int(w * 1.5)) if mode == 1 else (max(1, h // 2), max(1, w // 2)))
    seed = int(rng.integers(0, 10000))
    cls, logits = synth.panoptic_inputs(1, N, Np, ncls, Hm, Wm, seed)
    meta = dict(img_shape=(h, w, 3), batch_input_shape=(Hb, Wb), ori_shape=(Ho, Wo, 3))
    K = min(Np, Np * T)
    seg, info, nseg = vkn.ops.panoptic_joint(torch.from_numpy(cls).to(dev), torch.from_numpy(logits).to(dev), Np, T, K, 0.25, 0.6,
                                             (h, w), (Hb, Wb), (Ho, Wo), upsample_stride=up)
    with torch.no_grad():
        r = O.panoptic_joint(torch.from_numpy(cls)[0], torch.from_numpy(logits)[0], Np, T, K, 0.25, 0.6, meta, upsample_stride=up)
    tag = ('panoptic', Hm, Wm, up, (Hb, Wb), (h, w), (Ho, Wo), T, seed)
    info = info[0].cpu().numpy()
    assert int(nseg[0]) >= 0, tag
    assert np.array_equal(info[:, 0], r['rows'].numpy()) and np.array_equal(info[:, 1], r['total_labels'].numpy()), tag
    near = r['margin'].numpy() < 1e-6
    diff = seg[0].cpu().numpy() != r['panoptic_seg'].numpy()
    if np.array_equal(info[:, 2], r['seg_of'].numpy()):
        assert not (diff & ~near).any(), tag
    else:   # a segment decision flipped: only legitimate when an area ratio sits on the threshold because of near-tie pixels
        a, o = r['area'].numpy().astype(float), np.maximum(r['orig'].numpy().astype(float), 1)
        assert near.sum() > 0 and (np.abs(a / o - 0.6) < (2 * near.sum() + 2) / o).any(), tag


def t_head_handoff():
    from test_host_logic import _cfg
    C, N = 64, int(rng.integers(3, 150))
    H, W = int(rng.choice([8, 16, 24])), int(rng.choice([8, 16, 40]))
    B = int(rng.integers(1, 4))
    nth = 2
    head = vkn.build_head(_cfg(False, C=C, heads=8, ffn=128, ncls=5, n_thing=nth, n_stuff=3, S=2, up=2, nprop=max(1, N - 3)))
    head.init_weights()
    head = head.to(dev).eval()
    x, pf = torch.randn(B, C, H, W, device=dev), torch.randn(B, N, C, device=dev)
    mp = torch.randn(B, N, H, W, device=dev) * 3
    dims = head.mask_head[0].make_dims(B, N, H, W)
    packs = [h.stage_pack(torch.device(dev)) for h in head.mask_head]
    a = vkn.ops.head_forward(dims, packs, x, pf, mp, None, 2, flags=0)
    b = vkn.ops.head_forward(dims, packs, x, pf, mp, None, 2, flags=4)
    for u, v in zip(a[:4], b[:4]):
        assert torch.equal(u, v), ('handoff', B, N, H, W)


def t_assign():
    N, G, ncls = int(rng.integers(1, 128)), int(rng.integers(1, 60)), int(rng.integers(1, 9))
    H, W = int(rng.integers(2, 40)), int(rng.integers(2, 60))
    lo, cl, gt, lab = (torch.from_numpy(a) for a in synth.assign_inputs(N, G, ncls, H, W, int(rng.integers(0, 10000))))
    a = vkn.MaskHungarianAssigner(cls_cost=dict(type='FocalLossCost', weight=2.0), dice_cost=dict(type='DiceCost', weight=4.0, pred_act=True),
                                  mask_cost=dict(type='MaskCost', weight=1.0, pred_act=True))
    cost = a.cost_matrix(lo.to(dev), cl.to(dev), gt.to(dev), lab.to(dev))
    with torch.no_grad():
        want = O.assign_costs(lo, cl, gt, lab)
    assert float((cost.cpu() - want).abs().max()) < 5e-5, ('assign', N, G, ncls, H, W)


def t_kernel_init():
    B, Np, C = int(rng.integers(1, 3)), int(rng.integers(1, 120)), int(rng.choice([32, 64, 256]))
    H, W = int(rng.integers(1, 30)), int(rng.integers(2, 30))
    nth = int(rng.integers(0, 4))
    ncls = nth + int(rng.integers(1, 20))
    sem_on = bool(rng.integers(0, 2))
    loc = torch.randn(B, C, H, W)
    sem = torch.randn(B, C, H, W) if sem_on else None
    iw = torch.randn(Np, C, 1, 1) * 0.2
    sw = torch.randn(ncls, C, 1, 1) * 0.2 if sem_on else None
    sb = torch.randn(ncls) if sem_on else None
    cat = sem_on and bool(rng.integers(0, 2))
    d = lambda t: t.to(dev) if t is not None else None  # noqa: E731
    prop, xf, masks, seg = vkn.ops.kernel_init(d(loc), d(sem), d(iw), d(sw), d(sb), nth, cat, True)
    rp, rx, rm, rs = O.kernel_init(iw, loc, sem, sw, sb, nth, cat)
    tag = ('kernel_init', B, Np, C, H, W, nth, ncls, sem_on, cat)
    assert float((masks.cpu() - rm).abs().max()) < 1e-4, tag
    if sem_on:
        assert torch.equal(xf.cpu(), rx), tag
    bits = (masks[:, :Np].cpu() >= vkn.ops.thr_logit(0.5)).double()
    want = iw.reshape(1, Np, C).double() + torch.einsum('bnhw,bchw->bnc', bits, xf.cpu().double())
    assert float((prop[:, :Np].cpu().double() - want).abs().max()) < 1e-4 * max(1.0, float(want.abs().max())), tag


def t_head_c256():
    from test_host_logic import _cfg
    N = int(rng.integers(2, 170))
    B = int(rng.integers(1, 5))
    H, W = int(rng.choice([4, 8, 12])), int(rng.choice([8, 16]))
    video = bool(rng.integers(0, 2))
    key = ('h256', video)
    if key not in _heads:
        h = vkn.build_head(_cfg(video, C=256, heads=8, ffn=2048, ncls=19, n_thing=2, n_stuff=17, S=1, up=2, nprop=100))
        h.init_weights()
        _heads[key] = h.to(dev).eval()
    head = _heads[key]
    x, pf = torch.randn(B, 256, H, W, device=dev), torch.randn(B, N, 256, device=dev)
    mp = torch.randn(B, N, H, W, device=dev) * 3
    first = torch.randn(1, N, 256, device=dev)
    dims = head.mask_head[0].make_dims(B, N, H, W)
    packs = [h.stage_pack(torch.device(dev)) for h in head.mask_head]
    kw = dict(clip_first_prev=first) if video else {}
    a = vkn.ops.head_forward(dims, packs, x, pf, mp, None, 2, **kw)                 # bf16x3 GEMMs, fused FFN, composite weights
    e = vkn.ops.head_forward(dims, packs, x, pf, mp, None, 2, flags=2, **kw)        # exact-fp32 GEMMs, plain chain
    tag = ('head256', video, B, N, H, W)
    # one stage (no intermediate binarisation): the two GEMM paths must agree to fp32 rounding on every output
    assert float((a[1] - e[1]).abs().max()) < 1e-5, tag                                                   # cls probabilities
    assert float((a[0] - e[0]).abs().max()) < 1e-4 * max(1.0, float(e[0].abs().max())), tag              # kernels
    assert float((a[2] - e[2]).abs().max()) < 1e-3 * max(1.0, float(e[2].abs().max()) / 50), tag          # mask logits
    if video:
        assert float((a[4] - e[4]).abs().max()) < 1e-4 * max(1.0, float(e[4].abs().max())), tag


def t_chain_persistent():
    """The persistent row-owner chain (k_chain_a / k_chain_c, forced) against the launch-per-GEMM chain and the exact-fp32 GEMM chain:
    random row counts (ragged last tile, 1 .. 20 row tiles), hidden widths 256 .. 2048, class counts 1 .. 200, S = 1..3 stages, with /
    without the video link, whole-head calls (so the fused hand-off and the weight warm-up in the gather reduction are in the loop)."""
    from test_host_logic import _cfg
    N = int(rng.integers(2, 200))
    B = int(rng.integers(1, 5))
    H, W = int(rng.choice([4, 8])), int(rng.choice([8, 16]))
    video = bool(rng.integers(0, 2))
    ff = int(rng.choice([256, 512, 1024, 2048]))
    ncls = int(rng.choice([1, 3, 19, 40, 124, 200]))
    S = int(rng.integers(1, 4))
    key = ('chain', video, ff, ncls, S)
    if key not in _heads:
        n_stuff = 1 if ncls < 3 else 2
        h = vkn.build_head(_cfg(video, C=256, heads=8, ffn=ff, ncls=ncls, n_thing=max(ncls - n_stuff, 0) or 1, n_stuff=n_stuff if ncls > 1 else 0,
                                S=S, up=2, nprop=100))
        h.init_weights()
        _heads[key] = h.to(dev).eval()
    head = _heads[key]
    x, pf = torch.randn(B, 256, H, W, device=dev), torch.randn(B, N, 256, device=dev)
    mp = torch.randn(B, N, H, W, device=dev) * 3
    first = torch.randn(1, N, 256, device=dev)
    dims = head.mask_head[0].make_dims(B, N, H, W)
    packs = [h.stage_pack(torch.device(dev)) for h in head.mask_head]
    kw = dict(clip_first_prev=first) if video else {}
    tag = ('chain', video, B, N, H, W, ff, ncls, S)
    LAST['tag'] = tag
    p = vkn.ops.head_forward(dims, packs, x, pf, mp, None, 2, flags=vkn.ops.FLAG_CHAIN_PERSISTENT, **kw)
    p2 = vkn.ops.head_forward(dims, packs, x, pf, mp, None, 2, flags=vkn.ops.FLAG_CHAIN_PERSISTENT, **kw)
    assert all(a is None or torch.equal(a, b) for a, b in zip(p, p2)), ('not deterministic',) + tag
    assert all(a is None or bool(torch.isfinite(a).all()) for a in p), ('non-finite',) + tag
    # round 5: the few-row chain (vkn_ksplit.hip: column-spread phases, LayerNorm in the consumer, few-row link) — deterministic, finite
    f = vkn.ops.head_forward(dims, packs, x, pf, mp, None, 2, flags=vkn.ops.FLAG_CHAIN_KSPLIT, **kw)
    f2 = vkn.ops.head_forward(dims, packs, x, pf, mp, None, 2, flags=vkn.ops.FLAG_CHAIN_KSPLIT, **kw)
    assert all(a is None or torch.equal(a, b) for a, b in zip(f, f2)), ('few-row chain not deterministic',) + tag
    assert all(a is None or bool(torch.isfinite(a).all()) for a in f), ('few-row chain non-finite',) + tag
    if S == 1:      # one stage: no binarisation between the chains -> fp32-rounding agreement on every output
        q = vkn.ops.head_forward(dims, packs, x, pf, mp, None, 2, flags=vkn.ops.FLAG_CHAIN_LAUNCHES, **kw)
        e = vkn.ops.head_forward(dims, packs, x, pf, mp, None, 2, flags=2, **kw)
        for other in (q, e, f):
            assert float((p[1] - other[1]).abs().max()) < 1e-5, tag
            assert float((p[0] - other[0]).abs().max()) < 1e-4 * max(1.0, float(other[0].abs().max())), tag
            assert float((p[2] - other[2]).abs().max()) < 1e-3 * max(1.0, float(other[2].abs().max()) / 50), tag
            if video:
                assert float((p[4] - other[4]).abs().max()) < 1e-4 * max(1.0, float(other[4].abs().max())), tag
    vkn.ops.workspace_status()


def t_head_fused():
    """fused decode->gather pass vs bit words vs fp32 logits between stages, side-stream vs serial link: bit-identical outputs."""
    from test_host_logic import _cfg
    C = int(rng.choice([64, 128, 256]))
    N = int(rng.integers(3, 257))                                      # (> 128: two row chunks in one fused launch)
    H, W = int(rng.choice([8, 16, 32])), int(rng.choice([8, 16, 64]))   # H * W % 64 == 0: the fused pass is eligible
    B = int(rng.integers(1, 6))
    S = int(rng.integers(2, 4))
    video = bool(rng.integers(0, 2))
    key = ('fused', C, S, video)
    if key not in _heads:
        h = vkn.build_head(_cfg(video, C=C, heads=8, ffn=256 if C < 256 else 2048, ncls=7, n_thing=2, n_stuff=5, S=S, up=2, nprop=50))
        h.init_weights()
        _heads[key] = h.to(dev).eval()
    head = _heads[key]
    x, pf = torch.randn(B, C, H, W, device=dev), torch.randn(B, N, C, device=dev)
    mp = torch.randn(B, N, H, W, device=dev) * 3
    first = torch.randn(1, N, C, device=dev)
    dims = head.mask_head[0].make_dims(B, N, H, W)
    packs = [h.stage_pack(torch.device(dev)) for h in head.mask_head]
    kw = dict(clip_first_prev=first) if video else {}
    ref = vkn.ops.head_forward(dims, packs, x, pf, mp, None, 2, flags=4 | 32, **kw)      # fp32 logits hand-off, serial link
    for fl in (0, 16, 32):
        out = vkn.ops.head_forward(dims, packs, x, pf, mp, None, 2, flags=fl, **kw)
        for u, v in zip(out, ref):
            assert (u is None and v is None) or torch.equal(u, v), ('fused', fl, C, S, video, B, N, H, W)


def t_xhalf():
    """fp16 / bf16 storage of x: same bits as the fp32 kernels on the rounded x (gather, decode, fused pass, head)."""
    from test_host_logic import _cfg
    C = int(rng.choice([64, 128, 256]))
    N = int(rng.integers(1, 200))
    H, W = int(rng.choice([8, 16, 24])), int(rng.choice([8, 16, 40]))
    if (H * W) % 64:
        W = 16
    B = int(rng.integers(1, 5))
    dt = torch.float16 if rng.integers(0, 2) else torch.bfloat16
    x = torch.randn(B, C, H, W, device=dev)
    x = torch.where(x.abs() < 2.0 ** -13, torch.full_like(x, 2.0 ** -13), x)      # bf16 -> f16 exact in the normal range
    xh = x.to(dt)
    xr = xh.float()
    m = torch.randn(B, N, H, W, device=dev) * 3
    k = torch.randn(B, N, C, device=dev) * 0.2
    kb = torch.randn(B, N, device=dev)
    hi, lo = vkn.ops.split_planes(k)
    tag = ('xhalf', str(dt), B, N, C, H, W)
    a, b = vkn.ops.mask_gather(xh, m), vkn.ops.mask_gather(xr, m)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]), tag
    assert torch.equal(vkn.ops.mask_decode_planes(xh, hi, lo, N, kb), vkn.ops.mask_decode_planes(xr, hi, lo, N, kb)), tag
    if vkn._lib.lib().vkn_decode_gather_supported(C, H * W):
        a, b = vkn.ops.decode_gather(xh, hi, lo, N, kb), vkn.ops.decode_gather(xr, hi, lo, N, kb)
        assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]), tag


def t_lsap():
    """device LSAP (one wavefront per matrix) vs the host solver (== scipy): identical pairs on random / tied matrices."""
    mats = []
    for _ in range(48):
        nr, nc = int(rng.integers(1, 129)), int(rng.integers(1, 257))
        kind = int(rng.integers(0, 3))
        m = (rng.integers(0, 5, (nr, nc)).astype(np.float32) if kind == 0 else
             rng.standard_normal((nr, nc)).astype(np.float32) if kind == 1 else (rng.standard_normal((nr, nc)) * 2).round(1).astype(np.float32))
        mats.append(m)
    gts, rows, cols, st = vkn.ops.lsap_device([torch.from_numpy(m).to(dev) for m in mats])
    assert not bool(st.any())
    for m, g, r, c in zip(mats, gts, rows, cols):
        hr, hc = vkn.ops.lsap(m)
        assert np.array_equal(r.cpu().numpy(), hr) and np.array_equal(c.cpu().numpy(), hc), ('lsap', m.shape)
        want = np.zeros(m.shape[0], dtype=np.int64)
        want[hr] = hc + 1
        assert np.array_equal(g.cpu().numpy(), want), ('lsap gt_inds', m.shape)


def t_tracker():
    """device quasi-dense tracker vs the flat-table oracle on random videos: survivors, labels, ids of every frame."""
    from oracle.tracker_oracle import TrackerOracle, random_video
    metric = str(rng.choice(['bisoftmax', 'softmax', 'cosine']))
    cfg = dict(init_score_thr=0.5, obj_score_thr=0.35, match_score_thr=0.5, memo_tracklet_frames=int(rng.integers(2, 6)),
               memo_backdrop_frames=int(rng.integers(0, 4)), memo_momentum=0.8, nms_conf_thr=0.5, nms_backdrop_iou_thr=0.3,
               nms_class_iou_thr=0.7, with_cats=bool(rng.integers(0, 2)), match_metric=metric)
    emb = int(rng.choice([32, 64, 256]))
    trk = vkn.build_tracker(dict(cfg, type='QuasiDenseEmbedTracker', max_tracklets=1024))
    ora = TrackerOracle(**cfg)
    for t, (bb, lab, em) in enumerate(random_video(8, int(rng.integers(5, 100)), emb, 3, int(rng.integers(0, 1 << 30)))):
        b, l_, ids = trk.match(torch.from_numpy(bb).to(dev), torch.from_numpy(lab).to(dev), torch.from_numpy(em).to(dev), t)
        rb, rl, rids = ora.step(torch.from_numpy(bb), torch.from_numpy(lab), torch.from_numpy(em), t)
        assert np.array_equal(b.cpu().numpy(), rb.numpy()) and np.array_equal(l_.cpu().numpy(), rl.numpy()), ('tracker', metric, t)
        assert np.array_equal(ids.cpu().numpy(), rids.numpy()), ('tracker ids', metric, t)


def t_link_heads():
    """previous_link / previous_type video heads: one in-call clip == frame-by-frame calls, bit for bit (frame-sequential last stage)."""
    plink = [None, 'update_dynamic_cov', 'link_atten'][int(rng.integers(0, 3))]
    ptype = ['ffn', 'update', 'update_obj'][int(rng.integers(0, 3))]
    C = int(rng.choice([64, 128]))
    key = ('link', C, plink, ptype)
    if key not in _heads:
        cfgd = vkn.configs.roi_head_cfg(True, C=C, heads=8, ffn=128, ncls=5, n_thing=2, n_stuff=3, S=2, up=2, nprop=20,
                                        mask_over=dict(previous_link=plink, previous_type=ptype))
        h = vkn.build_head(cfgd)
        h.init_weights()
        _heads[key] = h.to(dev).eval()
    head = _heads[key]
    T, N = int(rng.integers(2, 5)), 23
    H, W = int(rng.choice([8, 16])), int(rng.choice([8, 16]))
    x, pf = torch.randn(T, C, H, W, device=dev), torch.randn(T, N, C, 1, 1, device=dev)
    mp = torch.randn(T, N, H, W, device=dev) * 3
    first = torch.randn(1, N, C, 1, 1, device=dev)
    clip = head.clip_forward(x, pf, mp, first)
    prev = first
    for t in range(T):
        one = head.clip_forward(x[t:t + 1], pf[t:t + 1], mp[t:t + 1], prev)
        for u, v in zip(one, clip):
            if torch.is_tensor(u):
                assert torch.equal(u[0], v[t]), ('link heads', plink, ptype, t)
        prev = one[0][0:1]


def t_query_merge():
    """clip-level attention query merge (<= 256 keys: LDS-staged attention; more: k_attn_long) vs the oracle's restatement."""
    C = int(rng.choice([64, 128, 256]))
    B, N, Fr = int(rng.integers(1, 4)), int(rng.integers(4, 120)), int(rng.integers(1, 7))
    with_pos = bool(rng.integers(0, 2))
    key = ('qm', C)
    if key not in _heads:
        shapes = {'query_merge_attn.attn.in_proj_weight': (3 * C, C), 'query_merge_attn.attn.in_proj_bias': (3 * C,),
                  'query_merge_attn.attn.out_proj.weight': (C, C), 'query_merge_attn.attn.out_proj.bias': (C,),
                  'query_merge_norm.weight': (C,), 'query_merge_norm.bias': (C,),
                  'query_merge_ffn.layers.0.0.weight': (8 * C, C), 'query_merge_ffn.layers.0.0.bias': (8 * C,),
                  'query_merge_ffn.layers.1.weight': (C, 8 * C), 'query_merge_ffn.layers.1.bias': (C,),
                  'query_merge_ffn_norm.weight': (C,), 'query_merge_ffn_norm.bias': (C,)}
        sd = {k: torch.from_numpy(v) for k, v in synth.state_dict_like(shapes, 5 + C).items()}
        named = {k: v.to(dev) for k, v in sd.items()}
        _heads[key] = (sd, vkn.ops.link_pack(named, torch.device(dev), None, 'query_merge_attn', 'query_merge_norm', 'query_merge_ffn',
                                             'query_merge_ffn_norm'), named)
    sd, pack, _ = _heads[key]
    q, k = torch.randn(B, N, C), torch.randn(B, Fr * N, C)
    pos = torch.randn(N, C) if with_pos else None
    ref = O.query_merge(sd, '', q, k, pos)
    out = vkn.ops.query_merge(vkn.ops.make_dims(B, N, C, 8, 8, 8, 8 * C, 1, 0, 0), pack, q.to(dev), k.to(dev), pos.to(dev) if with_pos else None)
    assert float((out.cpu() - ref).abs().max()) < 3e-4, ('query_merge', B, N, C, Fr, with_pos)


def t_train_ops():
    """round-3 training kernels against torch: focal loss (value + gradient, per-row / per-element weights), the upsample adjoint
    (every scale the entry takes, the S = 2 lane-exchange kernel when W % 64 == 0), the fused mask losses' gradient."""
    with torch.enable_grad():
        M, ncls = int(rng.integers(1, 700)), int(rng.integers(1, 140))
        z = (torch.randn(M, ncls, device=dev) * 4)
        lab = torch.randint(0, ncls + 1, (M,), device=dev)
        wk = int(rng.integers(0, 3))
        w = None if wk == 0 else (torch.rand((M,) if wk == 1 else (M, ncls), device=dev) > 0.3).float()
        avg = torch.tensor(float(rng.integers(1, 50)), device=dev)
        loss = vkn.losses.FocalLoss(use_sigmoid=True, gamma=2.0, alpha=0.25, loss_weight=float(rng.uniform(0.5, 3)))
        za, zb = z.clone().requires_grad_(True), z.clone().requires_grad_(True)
        la = loss(za, lab, w, avg_factor=avg)
        loss.fused = False
        lb = loss(zb, lab, w, avg_factor=avg)
        la.backward(); lb.backward()
        assert abs(float(la) - float(lb)) <= 2e-5 * max(abs(float(lb)), 1e-6), ('focal', M, ncls, wk)
        assert float((za.grad - zb.grad).abs().max()) <= 2e-5 * float(zb.grad.abs().max()) + 1e-9, ('focal grad', M, ncls, wk)
        S = int(rng.choice([1, 2, 2, 2, 3, 4, 8]))
        H, W = int(rng.integers(1, 24)), int(rng.choice([int(rng.integers(1, 80)), 64, 128, 192]))
        x = torch.randn(int(rng.integers(1, 3)), int(rng.integers(1, 5)), H, W, device=dev)
        g = torch.randn(x.shape[0], x.shape[1], H * S, W * S, device=dev)
        xa, xb = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
        vkn.autograd.upsample_bilinear(xa, S).backward(g)
        F.interpolate(xb, scale_factor=S, mode='bilinear', align_corners=False).backward(g)
        assert float((xa.grad - xb.grad).abs().max()) < 2e-5 * float(xb.grad.abs().max()), ('upsample adjoint', S, tuple(x.shape))


def t_attn_widths():
    """one stage through k_attn_mfma for head widths 16 / 32 / 64 and 1 .. 8 key blocks (N <= 256) against the oracle."""
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    from test_host_logic import _cfg
    from helpers import make_case
    C, heads = [(256, 4), (256, 8), (128, 8), (256, 16), (128, 4), (64, 4)][int(rng.integers(0, 6))]
    N = int(rng.integers(13, 250))
    kw = dict(C=C, heads=heads, ffn=2 * C, ncls=19, n_thing=8, n_stuff=11, S=1, up=1, nprop=N - 11)
    case = dict(kw, N=N, H=4, W=8, B=int(rng.integers(1, 3)), seed=int(rng.integers(0, 1 << 20)), video=0)
    key = ('aw', C, heads, N)
    head = vkn.build_head(_cfg(False, **kw))
    cfg, sd, x, pf, mp, _ = make_case(case)
    head.load_state_dict(sd, strict=True)
    head = head.to(dev).eval()
    traces = []
    O.iter_head_mask_preds(sd, x, pf, mp, cfg, traces=traces)
    B = case['B']
    dims = head.mask_head[0].make_dims(B, N, 4, 8)
    pack = head.mask_head[0].stage_pack(torch.device(dev))
    _, _, o0, _, _ = vkn.ops.stage_forward(dims, pack, x.to(dev), pf.reshape(B, N, C).to(dev), mp.to(dev))
    assert float((o0.cpu() - traces[0]['obj_feat'].reshape(B, N, C)).abs().max()) < 3e-4, ('attention widths', C, heads, N, B)


def t_chain_train():
    """round-4 training-chain kernels (csrc/vkn_train.hip) at random shapes against torch fp64 autograd: Linear (forward, dA on the
    transposed images incl. K = 512 / 768 and ragged out-feature counts, dW / db), LayerNorm + activation + residual (row and column
    workgroups), the attention core (packed and cross, every head width, ragged key / query blocks: matrix-core and VALU backward)."""
    ct = vkn.chain_train

    def rel(a, b):
        return float((a.double() - b).abs().max()) / max(float(b.abs().max()), 1e-30)

    with torch.enable_grad():
        M = int(rng.integers(1, 700))
        K = 32 * int(rng.integers(1, 25)) if rng.random() < 0.7 else 2048
        Nout = int(rng.choice([int(rng.integers(1, 300)), 256, 512, 768, 2048]))
        act, bias = int(rng.integers(0, 2)), bool(rng.integers(0, 2))
        a = torch.randn(M, K, device=dev).requires_grad_(True)
        w = (torch.randn(Nout, K, device=dev) * 0.05).requires_grad_(True)
        b = (torch.randn(Nout, device=dev) * 0.1).requires_grad_(True) if bias else None
        gy = torch.randn(M, Nout, device=dev) * 1e-2
        y = ct.linear(a, w, b, act=act)
        y.backward(gy)
        ad, wd = a.detach().double().requires_grad_(True), w.detach().double().requires_grad_(True)
        bd = b.detach().double().requires_grad_(True) if bias else None
        yr = F.linear(ad, wd, bd)
        yr = torch.relu(yr) if act else yr
        yr.backward(gy.double())
        tag = ('linear', M, K, Nout, act, bias)
        assert rel(y.detach(), yr.detach()) < 2e-5, tag
        # (a ReLU within fp32 rounding of zero flips in fp32: compare the gradients where the two forwards agree on the sign)
        if not act or bool(((y.detach() > 0) == (yr.detach() > 0)).all()):
            assert rel(a.grad, ad.grad) < 5e-5 and rel(w.grad, wd.grad) < 5e-5, tag
            assert (not bias) or rel(b.grad, bd.grad) < 5e-5, tag

        M, C = int(rng.integers(1, 700)), 32 * int(rng.integers(1, 9))
        act, resid, sliced = int(rng.integers(0, 3)), bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
        wide = (torch.randn(M, 2 * C, device=dev) * 2).requires_grad_(True)
        x = wide[:, C:] if sliced else wide[:, :C].contiguous()
        r = torch.randn(M, C, device=dev).requires_grad_(True) if resid else None
        ln = torch.nn.LayerNorm(C).to(dev)
        with torch.no_grad():
            ln.weight.copy_(torch.randn(C, device=dev) * 0.5 + 1.0)
            ln.bias.copy_(torch.randn(C, device=dev) * 0.3)
        gy = torch.randn(M, C, device=dev) * 1e-2
        y = ct.layernorm(x, ln, act=act, resid=r)
        y.backward(gy)
        wdd = wide.detach().double().requires_grad_(True)
        xd = wdd[:, C:] if sliced else wdd[:, :C]
        rd = r.detach().double().requires_grad_(True) if resid else None
        gd, bd = ln.weight.detach().double().requires_grad_(True), ln.bias.detach().double().requires_grad_(True)
        zz = F.layer_norm(xd + rd if resid else xd, (C,), gd, bd, ln.eps)
        yr = torch.relu(zz) if act == 1 else torch.sigmoid(zz) if act == 2 else zz
        yr.backward(gy.double())
        tag = ('layernorm', M, C, act, resid, sliced)
        assert rel(y.detach(), yr.detach()) < 1e-5, tag
        if act != 1 or bool(((y.detach() > 0) == (yr.detach() > 0)).all()):
            assert rel(wide.grad, wdd.grad) < 5e-5 and rel(ln.weight.grad, gd.grad) < 5e-5 and rel(ln.bias.grad, bd.grad) < 5e-5, tag
            assert (not resid) or rel(r.grad, rd.grad) < 5e-5, tag

        hd = int(rng.choice([4, 8, 16, 32, 64]))
        heads = int(rng.choice([h for h in (1, 2, 4, 8) if h * hd <= 256]))
        C, B = heads * hd, int(rng.integers(1, 4))
        Nq, Nk = int(rng.integers(1, 200)), int(rng.integers(1, 200))
        packed = bool(rng.integers(0, 2))
        if packed:
            Nk = Nq
            q = (torch.randn(B * Nq, 3 * C, device=dev) * 0.7).requires_grad_(True)
            kv = None
        else:
            q = (torch.randn(B * Nq, C, device=dev) * 0.7).requires_grad_(True)
            kv = (torch.randn(B * Nk, 2 * C, device=dev) * 0.7).requires_grad_(True)
        go = torch.randn(B * Nq, C, device=dev) * 1e-2
        o = ct.attention(q, kv, B, heads)
        o.backward(go)
        qd = q.detach().double().requires_grad_(True)
        kvd = kv.detach().double().requires_grad_(True) if kv is not None else None
        qq, kk, vv = (qd[:, :C], qd[:, C:2 * C], qd[:, 2 * C:]) if packed else (qd, kvd[:, :C], kvd[:, C:])
        qh, kh, vh = (t.reshape(B, -1, heads, hd).transpose(1, 2) for t in (qq, kk, vv))
        orf = (torch.softmax(qh @ kh.transpose(-1, -2) / hd ** 0.5, -1) @ vh).transpose(1, 2).reshape(B * Nq, C)
        orf.backward(go.double())
        tag = ('attention', B, Nq, Nk, heads, hd, packed)
        assert rel(o.detach(), orf.detach()) < 2e-5, tag
        if Nk == 1:     # one key: P = 1, dS = 0 — the exact gradients of q and k are ZERO; ours are rounding noise of dO . (v - O)
            assert float(q.grad[:, :C].abs().max()) < 1e-6 and bool(torch.isfinite(q.grad).all()), tag
        else:
            assert rel(q.grad, qd.grad) < 5e-5 and (packed or rel(kv.grad, kvd.grad) < 5e-5), tag


def t_lowres_tail():
    """round 6: the training tail's low-res kernels at random shapes against the up-scaled forms they replace — assignment costs
    (vkn_assign_costs_lowres_batch_f32 vs fp64 on the up-scaled logits), forward sums and backward of the mask losses
    (vkn_mask_losses_{fwd,bwd}_lowres_f32 vs upsample + _bank kernels + the upsample adjoint)."""
    import ctypes
    L, ops = vkn._lib.lib(), vkn.ops
    S = int(rng.choice([2, 4]))
    h, w = int(rng.integers(1, 40)), int(rng.integers(1, 50))
    B, Ns, K = int(rng.integers(1, 4)), int(rng.integers(2, 140)), 0
    H, W = S * h, S * w
    P = H * W
    seed = int(rng.integers(0, 100000))
    LAST['tag'] = ('lowres tail', B, Ns, h, w, S, seed)
    g = torch.Generator().manual_seed(seed)
    low = (torch.randn(B, Ns, h, w, generator=g) * float(rng.choice([0.5, 3.0, 12.0]))).to(dev)
    K = int(rng.integers(1, min(B * Ns, 40) + 1))
    bank = (torch.rand(K, H, W, generator=g) > 0.6).float().to(dev)
    rowk = torch.full((B * Ns,), -1, dtype=torch.int32)
    tgt = torch.zeros(B * Ns, dtype=torch.int32)
    pos = torch.randperm(B * Ns, generator=g)[:K].sort()[0]
    rowk[pos] = torch.arange(K, dtype=torch.int32)
    tgt[pos] = torch.arange(K, dtype=torch.int32)
    rowk, tgt, posd = rowk.to(dev), tgt.to(dev), pos.to(dev)
    p = lambda t: ctypes.c_void_p(t.data_ptr())   # noqa: E731
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    scaled = ops.upsample_bilinear(low, S)
    ncl = L.vkn_mask_losses_lowres_chunks(h, w)
    rp1, rk1 = torch.zeros(K, ncl, 4, device=dev), torch.zeros(B, ncl, device=dev)
    lse1, top1 = torch.full((B, P), float('nan'), device=dev), torch.full((B, P), -7, dtype=torch.int32, device=dev)
    assert L.vkn_mask_losses_fwd_lowres_f32(p(low), p(bank), p(tgt), p(rowk), K, B, Ns, h, w, S, 1, p(rp1), p(lse1), p(top1), p(rk1), st) == 0
    z = scaled.double().reshape(B, Ns, P)
    assert float((torch.logsumexp(z, 1) - lse1.double()).abs().max()) < 2e-5 * max(1.0, float(z.abs().max()))
    if P % 4 == 0:
        nch, nbl = L.vkn_mask_losses_chunks(P), L.vkn_mask_losses_blocks(P)
        rp0, rk0 = torch.zeros(K, nch, 4, device=dev), torch.zeros(B, nbl, device=dev)
        lse0, top0 = torch.zeros(B, P, device=dev), torch.zeros(B, P, dtype=torch.int32, device=dev)
        assert L.vkn_mask_losses_fwd_bank_f32(p(scaled), p(bank), p(tgt), p(posd), p(rowk), K, B, Ns, P, 1, p(rp0), p(lse0), p(top0), p(rk0), st) == 0
        a, b_ = rp0.double().sum(1), rp1.double().sum(1)
        assert float(((a - b_).abs() / a.abs().clamp(min=1.0)).max()) < 3e-6 and torch.equal(top0, top1)
        a_, bc = (torch.rand(K, generator=g) * 50).to(dev), (torch.rand(K, generator=g) * 50 + 60).to(dev)
        one = torch.ones(1, device=dev)
        out_lr, gs = torch.full_like(low, float('nan')), torch.empty_like(scaled)
        assert L.vkn_mask_losses_bwd_lowres_f32(p(low), p(bank), p(tgt), p(rowk), p(a_), p(bc), p(one), p(one), p(one), 1.0, 4.0, 0.1, K,
                                                p(lse1), p(top1), B, Ns, h, w, S, 1, p(out_lr), st) == 0
        assert L.vkn_mask_losses_bwd_bank_f32(p(scaled), p(bank), p(tgt), p(rowk), p(a_), p(bc), p(one), p(one), p(one), 1.0, 4.0, 0.1, K,
                                              p(lse1), p(top1), B, Ns, P, 1, p(gs), st) == 0
        ref = ops.upsample_bilinear_bwd(gs, S)
        assert float((out_lr - ref).abs().max()) < 3e-6 * float(ref.abs().max())
    # assignment costs
    N = int(rng.integers(1, min(Ns, 256) + 1))
    Gs = [int(rng.integers(1, 45)) for _ in range(B)]
    if ops.assign_costs_lowres_supported(N, Gs, h, w, S):
        ncls = int(rng.integers(1, 9))
        gts = [(torch.rand(G, H, W, generator=g) > 0.7).float().to(dev) for G in Gs]
        if rng.random() < 0.5:
            gts = [F.avg_pool2d(t[None], 3, 1, 1)[0].contiguous() for t in gts]
        cls = [torch.randn(N, ncls, generator=g).to(dev) for _ in Gs]
        labs = [torch.randint(0, ncls, (G,), generator=g).to(dev) for G in Gs]
        got = ops.assign_costs_lowres_batch([low[b][:N] for b in range(B)], S, cls, gts, labs)
        for b in range(B):
            pz = scaled[b][:N].double().sigmoid()
            p1, p2, gd = pz.clamp(0.001, 1.0).flatten(1), pz.clamp(0.01, 1.0).flatten(1), gts[b].double().flatten(1)
            dice = -(2 * p1 @ gd.t()) / ((p1 * p1).sum(1, keepdim=True) + 1e-3 + (gd * gd).sum(1)[None] + 1e-3)
            mcost = -(p2 @ gd.t() + (1 - p2) @ (1 - gd).t()) / P
            pc = cls[b].double().sigmoid()
            foc = (-(pc + 1e-12).log() * 0.25 * (1 - pc) ** 2 + (1 - pc + 1e-12).log() * 0.75 * pc ** 2)[:, labs[b]]
            want = 2.0 * foc + 4.0 * dice + mcost
            assert float((got[b].double() - want).abs().max()) < 1e-5 * max(2.0, float(want.abs().max())), ('assign', b)


_heads = {}
only = sys.argv[2:]
with torch.no_grad():
    for name, fn in (('gather / decode', t_gather_decode), ('upsample', t_upsample), ('panoptic joint', t_panoptic),
                     ('head bit vs logits hand-off', t_head_handoff), ('assignment costs', t_assign),
                     ('kernel init', t_kernel_init), ('head C=256 split vs exact GEMMs', t_head_c256),
                     ('persistent chain vs launch chain vs exact', t_chain_persistent),
                     ('head fused / bits / logits / side stream', t_head_fused), ('half-storage x', t_xhalf),
                     ('device LSAP vs host solver', t_lsap), ('device tracker vs oracle', t_tracker),
                     ('link heads clip vs frame-by-frame', t_link_heads), ('VIS attention query merge', t_query_merge),
                     ('training ops vs torch', t_train_ops), ('attention head widths / key blocks', t_attn_widths),
                     ('training chain kernels vs torch fp64', t_chain_train), ('low-res training tail', t_lowres_tail)):
        if not only or any(o in name for o in only):
            section(name, fn)
print('soak: OK')

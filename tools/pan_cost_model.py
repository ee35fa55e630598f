#!/usr/bin/env python3
"""Cost model inputs of k_pan_argmax: survivors per output tile under the footprint-bound pruning, on the bench's
synthetic inputs and on the pan_* goldens' inputs — a torch / numpy EMULATION of the bound (the kernel's own arithmetic:
pan_coef / pan_region of csrc/vkn_panoptic.hip in float32), run on the CPU or the GPU.  usage: tools/pan_cost_model.py [bench|<golden> ...]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
TW, TH = 64, int(os.environ.get('PAN_TH', 16))


def coef(scale, dst, in_size):
    src = np.float32(scale) * (np.float32(dst) + np.float32(0.5)) - np.float32(0.5)
    src = max(src, np.float32(0.0))
    i0 = min(int(np.floor(src)), in_size - 1)
    return i0, i0 + (1 if i0 < in_size - 1 else 0)


def region(levels, o0, n):
    """[(in_size, scale)] from the logits up -> (origin, extent) of the logits region of outputs [o0, o0 + n)"""
    for in_size, scale in reversed(levels):
        a0, _ = coef(scale, o0, in_size)
        _, b1 = coef(scale, o0 + n - 1, in_size)
        o0, n = a0, b1 - a0 + 1
    return o0, n


def survivors(cls_sel, logits_sel, up, Hb, Wb, h, w, Ho, Wo):
    """cls_sel [K] scores, logits_sel [K, Hm, Wm] of the selected kernels of one frame -> [nty, ntx] survivor counts"""
    K, Hm, Wm = logits_sel.shape
    Ha, Wa = Hm * up, Wm * up
    ly = [(Hm, np.float32(1.0 / up)), (Ha, np.float32(Ha) / np.float32(Hb))]
    lx = [(Wm, np.float32(1.0 / up)), (Wa, np.float32(Wa) / np.float32(Wb))]
    if not (h == Ho and w == Wo):
        ly.append((h, np.float32(h) / np.float32(Ho)))
        lx.append((w, np.float32(w) / np.float32(Wo)))
    nty, ntx = (Ho + TH - 1) // TH, (Wo + TW - 1) // TW
    out = torch.zeros(nty, ntx, dtype=torch.int32)
    foot = []
    for ty in range(nty):
        y0, nh = region(ly, ty * TH, min(TH, Ho - ty * TH))
        rows = logits_sel[:, y0:y0 + nh]
        cmx, cmn = rows.amax(1), rows.amin(1)
        for tx in range(ntx):
            x0, nw = region(lx, tx * TW, min(TW, Wo - tx * TW))
            mx, mn = cmx[:, x0:x0 + nw].amax(1), cmn[:, x0:x0 + nw].amin(1)
            pm, pn = torch.sigmoid(mx), torch.sigmoid(mn)
            hi, lo = cls_sel * pm * (1 + 1e-5) + 1e-30, (cls_sel * pn * (1 - 1e-5)).clamp(min=0)
            out[ty, tx] = int(((hi >= lo.max()) | (pm * (1 + 1e-5) >= 0.5)).sum())
            foot.append(nw * nh)
    return out, float(np.mean(foot))


def report(tag, counts, foot, K):
    c = torch.cat([x.flatten() for x in counts]).float()
    print(f'{tag}: K = {K}, tiles {c.numel()}, footprint {foot:.1f} logits px per kernel and tile; survivors per tile: mean {c.mean():.2f}, '
          f'median {c.median():.0f}, p90 {c.quantile(0.9):.0f}, max {c.max():.0f}')


def select(cls, Np, T, Kt):
    """the reference's selection (thing top-k over (proposal, class), stuff diag sorted) -> (rows [K], scores [K])"""
    N, ncls = cls.shape
    th = cls[:Np, :T].flatten()
    sc, idx = th.topk(Kt)
    rows = idx // T
    st = torch.stack([cls[Np + j, T + j] for j in range(N - Np)]) if N > Np else cls.new_zeros(0)
    ss, si = st.sort(descending=True) if st.numel() else (st, st.long())
    return torch.cat([rows, Np + si]), torch.cat([sc, ss])


for name in (sys.argv[1:] or ['bench', 'pan_cfg', 'pan_kitti', 'pan_vipseg', 'pan_tiny']):
    if name == 'bench':
        import bench
        N, Np, Hm, Wm = 117, 100, 128, 256
        cls, logits = bench.panoptic_inputs(2, N, Np, 19, Hm, Wm, 'cpu')
        cs, foot = [], 0
        for b in range(2):
            rows, sc = select(cls[b], Np, 2, Np)
            c, foot = survivors(sc, logits[b][rows], 4, 1024, 2048, 1024, 2048, 1024, 2048)
            cs.append(c)
        report('bench.py panoptic_inputs (blobs + bands + noise), 1024x2048', cs, foot, len(sc))
    else:
        from helpers import load_pan_golden, make_pan_case
        g, case = load_pan_golden(name)
        cls, logits, meta = make_pan_case(case)
        cls, logits = torch.as_tensor(cls), torch.as_tensor(logits)
        cs, foot = [], 0
        img, bis, ori = meta['img_shape'][:2], meta['batch_input_shape'][:2], meta['ori_shape'][:2]
        for b in range(cls.shape[0]):
            rows, sc = select(cls[b], case['Np'], case['T'], case['Np'])
            c, foot = survivors(sc, logits[b][rows], case['up'], bis[0], bis[1], img[0], img[1], ori[0], ori[1])
            cs.append(c)
        report(f'golden {name} ({ori[0]}x{ori[1]})', cs, foot, len(sc))

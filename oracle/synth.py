"""Deterministic synthetic tensors for goldens and parity tests — TEST INFRASTRUCTURE ONLY.

Values come from an integer hash (splitmix64 finaliser) of the flat element index, so they are
bit-reproducible on any machine without relying on a framework RNG stream.  Used by
`oracle/gen_golden.py` (build container) and by `tests/` (build container and GPU box) to regenerate
the *inputs* of a golden case; only the reference's *outputs* are stored in `tests/golden/`.
"""
import math

import numpy as np

_M64 = (1 << 64) - 1


def _splitmix(idx: np.ndarray, salt: int) -> np.ndarray:
    with np.errstate(over='ignore'):
        z = idx.astype(np.uint64) + np.uint64((salt * 0x9E3779B97F4A7C15 + 0x632BE59BD9B4E019) & _M64)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return z


def uniform(shape, salt: int, lo=-1.0, hi=1.0) -> np.ndarray:
    """float32 tensor, U[lo,hi) from the hash of the flat index (24 random mantissa bits)."""
    n = int(np.prod(shape)) if len(shape) else 1
    z = _splitmix(np.arange(n, dtype=np.uint64), salt)
    u = (z >> np.uint64(40)).astype(np.float64) / float(1 << 24)
    return (lo + (hi - lo) * u).astype(np.float32).reshape(shape)


def normalish(shape, salt: int, std=1.0) -> np.ndarray:
    """Sum of four uniforms (Irwin-Hall), variance-normalised: smooth bell-shaped, bounded."""
    acc = np.zeros(shape, dtype=np.float64)
    for k in range(4):
        acc += uniform(shape, salt * 4 + k + 1000003, -1.0, 1.0).astype(np.float64)
    return (acc * (std / math.sqrt(4.0 / 3.0))).astype(np.float32)


def _key_salt(key: str, seed: int) -> int:
    h = 1469598103934665603
    for ch in key.encode():
        h = ((h ^ ch) * 1099511628211) & _M64
    return (h ^ (seed * 0x9E3779B97F4A7C15)) & 0x7FFFFFFF


def state_dict_like(shapes: dict, seed: int = 0) -> dict:
    """Weights for a {key: shape} map, mirroring the reference init *distribution*
    (knet/det/kernel_update_head.py:151-168: xavier-uniform on dim>1, fc_cls.bias = -log 99) but with
    non-trivial LayerNorm affine and biases so that every parameter influences the outputs."""
    out = {}
    for key, shape in shapes.items():
        shape = tuple(shape)
        salt = _key_salt(key, seed)
        if len(shape) > 1:
            rf = int(np.prod(shape[2:])) if len(shape) > 2 else 1
            fan_in, fan_out = shape[1] * rf, shape[0] * rf
            b = math.sqrt(6.0 / (fan_in + fan_out))
            out[key] = uniform(shape, salt, -b, b)
        elif key.endswith('weight'):
            # every 1-D `weight` on the path is a LayerNorm gain: ~ 1 +- 0.2
            out[key] = uniform(shape, salt, 0.8, 1.2)
        elif key.endswith('fc_cls.bias'):
            out[key] = (uniform(shape, salt, -0.2, 0.2) - math.log(99.0)).astype(np.float32)
        else:
            out[key] = uniform(shape, salt, -0.1, 0.1)
    return out


def head_inputs(B, N, C, H, W, seed=0, mask_scale=4.0):
    """x [B,C,H,W], proposal_feats [B,N,C,1,1], mask_preds [B,N,H,W]."""
    x = normalish((B, C, H, W), 11 + 7 * seed, 1.0)
    pf = normalish((B, N, C, 1, 1), 12 + 7 * seed, 1.0)
    mp = normalish((B, N, H, W), 13 + 7 * seed, mask_scale)
    return x, pf, mp


def panoptic_inputs(B, N, Np, ncls, Hm, Wm, seed):
    """Structured inputs of the post-head pipeline: cls probabilities [B,N,ncls] in (0,1) and low-res mask logits [B,N,Hm,Wm]
    that look like segmentation (one soft elliptic blob per thing kernel, one horizontal band per stuff kernel, plus noise), so
    that the joint merge accepts some kernels and rejects others (i.i.d. noise would reject everything)."""
    cls = uniform((B, N, ncls), 41 + 13 * seed, 0.02, 0.98)
    noise = normalish((B, N, Hm, Wm), 42 + 13 * seed, 0.7)
    cx = uniform((B, N), 43 + 13 * seed, 0.05, 0.95).astype(np.float64)
    cy = uniform((B, N), 44 + 13 * seed, 0.05, 0.95).astype(np.float64)
    r0 = float(np.sqrt(0.5 / max(Np, 1)))   # blobs shrink with the kernel count so that they overlap only partly
    rx = uniform((B, N), 45 + 13 * seed, 0.5 * r0, 1.5 * r0).astype(np.float64)
    ry = uniform((B, N), 46 + 13 * seed, 0.5 * r0, 1.5 * r0).astype(np.float64)
    ys = ((np.arange(Hm) + 0.5) / Hm)[None, None, :, None]
    xs = ((np.arange(Wm) + 0.5) / Wm)[None, None, None, :]
    d = np.sqrt(((xs - cx[..., None, None]) / rx[..., None, None]) ** 2 + ((ys - cy[..., None, None]) / ry[..., None, None]) ** 2)
    logits = np.clip(6.0 * (1.0 - d), -6.0, 6.0)
    ns = N - Np
    for j in range(ns):  # stuff: band j of ns
        inside = (ys >= j / ns) & (ys < (j + 1) / ns)
        logits[:, Np + j] = np.where(np.broadcast_to(inside[:, 0], (B, Hm, Wm)), 4.0, -4.0)
    return cls.astype(np.float32), (logits + noise).astype(np.float32)


def assign_inputs(N, G, ncls, H, W, seed):
    """Inputs of the train-time assignment for one image: mask logits [N,H,W] (blobs + noise), class logits [N,ncls],
    ground-truth masks [G,H,W] in {0,1} (G of the blobs, thresholded, so that a good matching exists) and labels [G]."""
    _, logits = panoptic_inputs(1, N, N, 1, H, W, 100 + seed)
    logits = logits[0]
    cls = normalish((N, ncls), 61 + 13 * seed, 2.0)
    pick = (uniform((G,), 62 + 13 * seed, 0.0, 1.0).astype(np.float64) * N).astype(np.int64)
    gt = (logits[pick] + normalish((G, H, W), 63 + 13 * seed, 1.0) > 0.5).astype(np.float32)
    labels = (uniform((G,), 64 + 13 * seed, 0.0, 1.0).astype(np.float64) * ncls).astype(np.int64)
    return logits.astype(np.float32), cls.astype(np.float32), gt, labels


def train_targets(B, n_thing, n_stuff, Hs, Ws, seed, gmin=2, gmax=5, soft=True):
    """Ground truth of one training batch at the resolution of the up-scaled mask predictions: per image a list entry with
    thing masks [G_i, Hs, Ws] (elliptic blobs; `soft` leaves bilinear-like soft borders, values in [0, 1], as the reference's
    down-sampled gt masks have — knet/det/knet.py:131), thing labels [G_i], and the stuff targets gt_sem_cls [S_i] (labels in
    [n_thing, n_thing + n_stuff)) with band masks gt_sem_seg [S_i, Hs, Ws]."""
    out = []
    ys = ((np.arange(Hs) + 0.5) / Hs)[None, :, None]
    xs = ((np.arange(Ws) + 0.5) / Ws)[None, None, :]
    for b in range(B):
        sd = 7001 + 131 * seed + 17 * b
        G = gmin + int(uniform((1,), sd, 0.0, 1.0)[0] * (gmax - gmin + 1) * 0.999)
        cx = uniform((G, 1, 1), sd + 1, 0.15, 0.85).astype(np.float64)
        cy = uniform((G, 1, 1), sd + 2, 0.15, 0.85).astype(np.float64)
        rx = uniform((G, 1, 1), sd + 3, 0.08, 0.3).astype(np.float64)
        ry = uniform((G, 1, 1), sd + 4, 0.08, 0.3).astype(np.float64)
        d = np.sqrt(((xs - cx) / rx) ** 2 + ((ys - cy) / ry) ** 2)
        gt = np.clip((1.0 - d) * (4.0 if soft else 1e6) + 0.5, 0.0, 1.0).astype(np.float32)
        labels = (uniform((G,), sd + 5, 0.0, 1.0).astype(np.float64) * n_thing).astype(np.int64)
        sem_cls, sem_seg = np.zeros((0,), np.int64), np.zeros((0, Hs, Ws), np.float32)
        if n_stuff > 0:
            present = uniform((n_stuff,), sd + 6, 0.0, 1.0) < 0.7
            present[0] = True
            idx = np.nonzero(present)[0]
            sem_cls = (idx + n_thing).astype(np.int64)
            sem_seg = np.stack([((ys[0] >= j / n_stuff) & (ys[0] < (j + 1) / n_stuff)).astype(np.float32).repeat(Ws, axis=1)
                                for j in idx])
        out.append(dict(gt_masks=gt, gt_labels=labels, gt_sem_cls=sem_cls, gt_sem_seg=sem_seg))
    return out


def clip_targets(num_clips, num_frames, n_cls, Hs, Ws, seed, gmin=2, gmax=4):
    """Ground truth of the clip-level VIS training (knet_vis: ref_gt_masks / ref_gt_labels / ref_gt_instance_ids): per clip a few
    instances (drifting soft elliptic blobs with one label each), every instance visible in a subset of the frames (always in at
    least one), listed per frame in a shuffled order.  Per clip: gt_masks = list over frames of [n_f, Hs, Ws] float32;
    gt_labels [M, 2] = (frame, label); gt_instance_ids [M, 2] = (frame, instance id) — the rows of a frame in its masks' order."""
    ys = ((np.arange(Hs) + 0.5) / Hs)[:, None]
    xs = ((np.arange(Ws) + 0.5) / Ws)[None, :]
    out = []
    for b in range(num_clips):
        sd = 9101 + 211 * seed + 29 * b
        G = gmin + int(uniform((1,), sd, 0.0, 1.0)[0] * (gmax - gmin + 1) * 0.999)
        ids = 3 + 2 * np.arange(G) + int(uniform((1,), sd + 1, 0.0, 1.0)[0] * 5)        # sparse, ascending instance ids
        labels = (uniform((G,), sd + 2, 0.0, 1.0).astype(np.float64) * n_cls).astype(np.int64)
        c0 = uniform((G, 2), sd + 3, 0.2, 0.8).astype(np.float64)
        vel = uniform((G, 2), sd + 4, -0.06, 0.06).astype(np.float64)
        rad = uniform((G, 2), sd + 5, 0.08, 0.28).astype(np.float64)
        vis = uniform((G, num_frames), sd + 6, 0.0, 1.0) < 0.75
        vis[np.arange(G), (uniform((G,), sd + 7, 0.0, 1.0).astype(np.float64) * num_frames).astype(np.int64)] = True
        gt_masks, lab_rows, id_rows = [], [], []
        for f in range(num_frames):
            present = np.nonzero(vis[:, f])[0]
            order = np.argsort(uniform((len(present),), sd + 40 + f, 0.0, 1.0), kind='stable')
            present = present[order]
            fm = []
            for g in present:
                cx, cy = c0[g] + vel[g] * f
                d = np.sqrt(((xs - cx) / rad[g, 0]) ** 2 + ((ys - cy) / rad[g, 1]) ** 2)
                fm.append(np.clip((1.0 - d) * 4.0 + 0.5, 0.0, 1.0).astype(np.float32))
                lab_rows.append((f, labels[g]))
                id_rows.append((f, ids[g]))
            gt_masks.append(np.stack(fm) if fm else np.zeros((0, Hs, Ws), np.float32))
        out.append(dict(gt_masks=gt_masks, gt_labels=np.array(lab_rows, dtype=np.int64).reshape(-1, 2),
                        gt_instance_ids=np.array(id_rows, dtype=np.int64).reshape(-1, 2)))
    return out


def tracker_sequence(T, n_obj, emb, n_cls, seed):
    """A synthetic video for the quasi-dense tracker: `n_obj` objects drift over T frames (some disappear / reappear), each frame
    lists detections in shuffled order: boxes [n,5] (x1,y1,x2,y2,score), labels [n], embeddings [n,emb] = object code + noise, plus
    a few low-score duplicates and clutter.  Returns a list of (bboxes, labels, embeds, true_object_index)."""
    codes = normalish((n_obj, emb), 900 + seed, 1.0) * 3.0
    cls = (uniform((n_obj,), 901 + seed, 0.0, 1.0).astype(np.float64) * n_cls).astype(np.int64)
    c0 = uniform((n_obj, 2), 902 + seed, 40.0, 600.0).astype(np.float64)
    vel = uniform((n_obj, 2), 903 + seed, -12.0, 12.0).astype(np.float64)
    size = uniform((n_obj, 2), 904 + seed, 20.0, 90.0).astype(np.float64)
    frames = []
    for t in range(T):
        vis = uniform((n_obj,), 910 + 31 * seed + t, 0.0, 1.0) < 0.85
        rows, labs, embs, who = [], [], [], []
        for o in (int(v) for v in np.nonzero(vis)[0]):
            c = c0[o] + vel[o] * t
            sc = float(uniform((1,), 920 + 97 * seed + 13 * t + o, 0.25, 0.99)[0])
            rows.append([c[0] - size[o, 0], c[1] - size[o, 1], c[0] + size[o, 0], c[1] + size[o, 1], sc])
            labs.append(cls[o])
            embs.append(codes[o] + normalish((emb,), 930 + 101 * seed + 17 * t + o, 0.3))
            who.append(o)
            if (o + t) % 4 == 0:      # a low-score near-duplicate of the same object
                rows.append([c[0] - size[o, 0] + 3, c[1] - size[o, 1] + 2, c[0] + size[o, 0] + 3, c[1] + size[o, 1] + 2, sc * 0.4])
                labs.append(cls[o])
                embs.append(codes[o] + normalish((emb,), 940 + 101 * seed + 17 * t + o, 0.3))
                who.append(o)
        order = np.argsort(uniform((len(rows),), 950 + seed * 7 + t, 0.0, 1.0))
        frames.append((np.asarray(rows, np.float32)[order], np.asarray(labs, np.int64)[order],
                       np.stack(embs).astype(np.float32)[order], np.asarray(who, np.int64)[order]))
    return frames

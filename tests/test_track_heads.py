"""`QuasiDenseMaskEmbedHeadGTMask` — the embedding head between the update head's tracking kernels and the quasi-dense tracker
 — against goldens captured from the reference's own class (oracle/gen_golden_tracker.py -> qd_embed_head.npz).
CPU: module tree, targets, similarities, losses (host-side torch).  GPU (-m gpu): the forward (HIP linear chain on pre-split weights)
and the hand-over to the tracker without a host hop."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle.embed_cases import EMBED_CASES, embed_case_inputs  # noqa: E402  (hash-formula inputs; no reference needed)

GOLDEN = os.path.join(ROOT, 'tests', 'golden', 'qd_embed_head.npz')
DEV = 'cuda:0'


def _head(vkn, cfg):
    return vkn.build_head(dict(cfg, type='QuasiDenseMaskEmbedHeadGTMask'))


@pytest.mark.parametrize('name', sorted(EMBED_CASES))
def test_embed_head_training_side_vs_reference_golden(vkn, name):
    """State-dict keys, `get_track_targets`, `match` (dot-product / temperature / cosine) and `loss` (MultiPosCrossEntropyLoss, L2Loss
    with margins and hard-negative mining) on the REFERENCE's embeddings: targets and weights bit-exact, similarities 1e-6, losses
    1e-5 relative."""
    g = np.load(GOLDEN)
    cfg, sizes, seed = EMBED_CASES[name]
    head = _head(vkn, cfg)
    sd, keys, refs, kres, rres, match = embed_case_inputs(cfg, sizes, seed)
    head.load_state_dict(sd, strict=True)
    assert sorted(head.state_dict()) == list(g[name + '_keys'])
    ke, re_ = torch.from_numpy(g[name + '_key_embeds']), torch.from_numpy(g[name + '_ref_embeds'])
    dists, cos = head.match(ke, re_, kres, rres)
    targets, weights = head.get_track_targets(match, kres, rres)
    for i in range(2):
        assert np.array_equal(targets[i].numpy(), g[f'{name}_targets{i}']) and np.array_equal(weights[i].numpy(), g[f'{name}_weights{i}'])
        assert np.max(np.abs(dists[i].numpy() - g[f'{name}_dists{i}'])) < 1e-5 * max(1.0, float(np.abs(g[f'{name}_dists{i}']).max()))
        if cos[i] is not None:
            assert np.max(np.abs(cos[i].numpy() - g[f'{name}_cos{i}'])) < 1e-6
        else:
            assert f'{name}_cos{i}' not in g.files
    losses = head.loss([torch.from_numpy(g[f'{name}_dists{i}']).clone() for i in range(2)],
                       [torch.from_numpy(g[f'{name}_cos{i}']).clone() if f'{name}_cos{i}' in g.files else None for i in range(2)],
                       [t.clone() for t in targets], [w.clone() for w in weights])
    assert sorted(losses) == sorted(k[len(name) + 1:] for k in g.files if k.startswith(name + '_loss_'))
    for k, v in losses.items():
        ref = float(g[f'{name}_{k}'])
        assert abs(float(v) - ref) < 1e-5 * max(1.0, abs(ref)), (k, float(v), ref)


def test_embed_head_has_no_cpu_fallback_and_rejects_conv_towers(vkn):
    cfg, _, _ = EMBED_CASES['emb_one']
    head = _head(vkn, cfg).eval()
    with pytest.raises(vkn.VknLibraryError):
        head(torch.zeros(3, cfg['in_channels'], 1, 1))
    with pytest.raises(NotImplementedError):
        _head(vkn, dict(cfg, num_convs=4, roi_feat_size=7))


@pytest.mark.gpu
@pytest.mark.parametrize('name', sorted(EMBED_CASES))
def test_embed_head_forward_vs_reference_golden(vkn, name):
    """`forward` on the GPU (one HIP GEMM launch per layer, bf16x3 split weights where K % 32 == 0) against the reference's
    embeddings; the autograd form (training) gives the same values and gradients flow to the parameters."""
    g = np.load(GOLDEN)
    cfg, sizes, seed = EMBED_CASES[name]
    head = _head(vkn, cfg)
    sd, keys, refs, *_ = embed_case_inputs(cfg, sizes, seed)
    head.load_state_dict(sd, strict=True)
    head = head.to(DEV).eval()
    with torch.no_grad():
        ke = head(torch.cat(keys, 0).to(DEV))
        re_ = head(torch.cat(refs, 0).to(DEV))
        assert head(torch.zeros(0, cfg['in_channels'], 1, 1, device=DEV)).shape == (0, cfg['embed_channels'])
    scale = max(1.0, float(np.abs(g[name + '_key_embeds']).max()))
    assert float((ke.cpu() - torch.from_numpy(g[name + '_key_embeds'])).abs().max()) < 2e-6 * scale
    assert float((re_.cpu() - torch.from_numpy(g[name + '_ref_embeds'])).abs().max()) < 2e-6 * scale
    head.train()
    xk = torch.cat(keys, 0).to(DEV).requires_grad_(True)
    out = head(xk)
    assert float((out.detach() - ke).abs().max()) < 1e-5 * scale
    out.square().sum().backward()
    assert xk.grad is not None and all(p.grad is not None for p in head.parameters())


@pytest.mark.gpu
def test_embeddings_feed_the_tracker_without_a_host_hop(vkn):
    """head -> embeddings -> `vkn_qd_tracker_match_f32`: device tensors all the way (what `simple_test` of the video detector does with
    `self.track_head` and `self.tracker`, knet/video/knet_quansi_dense_embed_fc_joint_train.py:560-590)."""
    cfg, sizes, seed = EMBED_CASES['emb_cfg']
    head = _head(vkn, cfg)
    head.load_state_dict(embed_case_inputs(cfg, sizes, seed)[0], strict=True)     # (fresh `init_weights` embeddings are ~1e-2: no match clears 0.5)
    head = head.to(DEV).eval()
    tracker = vkn.build_tracker(dict(type='QuasiDenseEmbedTracker', init_score_thr=0.35, obj_score_thr=0.3, match_score_thr=0.5,
                                     memo_tracklet_frames=5, memo_backdrop_frames=1, memo_momentum=0.8, nms_conf_thr=0.5,
                                     nms_backdrop_iou_thr=0.3, nms_class_iou_thr=0.7, with_cats=True, match_metric='bisoftmax'))
    gen = torch.Generator().manual_seed(0)
    feats = torch.randn(6, 256, 1, 1, generator=gen).to(DEV)
    boxes = torch.tensor([[10. * i, 5., 10. * i + 8., 30., 0.9 - 0.05 * i] for i in range(6)], device=DEV)
    labels = torch.zeros(6, dtype=torch.long, device=DEV)
    with torch.no_grad():
        ids0 = tracker.match(boxes, labels, head(feats), frame_id=0)[2]
        ids1 = tracker.match(boxes, labels, head(feats + 0.01 * torch.randn(6, 256, 1, 1, generator=gen).to(DEV)), frame_id=1)[2]
    assert ids0.tolist() == list(range(6)) and ids1.tolist() == ids0.tolist(), (ids0.tolist(), ids1.tolist())

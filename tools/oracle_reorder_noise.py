"""How much does the REFERENCE ALGORITHM ITSELF move when only its fp32 summation order changes?  (VERDICT r04 "close the parity
argument with measurements", item 5 (i).)

The free-running full-size parity tests bound flipped near-threshold mask bits and "clean" kernel rows of the HIP path against the
reference / the oracle (tests/test_gpu_parity.py: test_head_cfg5_cfg4_size_vs_reference_golden,
test_cfg2_size_free_running_vs_oracle_flip_budget).  This script measures the SAME quantities for the CPU oracle against itself, the
only difference between the two runs being the order in which fp32 sums are taken:
  * `perm k`   — the input channels of x (and, consistently, the input-channel axis of every stage's feat_transform weight) are
                 permuted: mathematically the identical function, the 1x1 conv sums its 256 channels in another order;
  * `threads t` — intra-op thread count t instead of the baseline's (changes the blocking of the BLAS / oneDNN reductions).
No GPU, no HIP code: this is the noise floor ANY re-implementation (another BLAS, another thread count) sits on.
   python tools/oracle_reorder_noise.py [--perms 4] [--out profiles/r05_oracle_reorder_noise.json]"""
import argparse
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from helpers import cfg_of, load_golden, make_case  # noqa: E402
import oracle.knet_oracle as O  # noqa: E402

THR = 8.940696716308594e-08   # smallest fp32 z with sigmoid(z) > 0.5 (ops.thr_logit(0.5))


def run(case, perm=None, threads=None):
    cfg, sd, x, pf, mp, prev = make_case(case)
    if perm is not None:
        x = x[:, perm].contiguous()
        sd = dict(sd)
        for k in list(sd):
            if k.endswith('feat_transform.conv.weight'):
                sd[k] = sd[k][:, perm].contiguous()
    old = torch.get_num_threads()
    if threads:
        torch.set_num_threads(threads)
    traces = []
    with torch.no_grad():
        out = O.iter_head_mask_preds(sd, x, pf, mp, cfg, previous_obj_feats=(prev if case.get('video') else None), traces=traces)
    torch.set_num_threads(old)
    return out, traces


def golden_metrics(name, out):
    """the quantities of test_head_cfg5_cfg4_size_vs_reference_golden, for an oracle run against the reference golden"""
    g, case = load_golden(name)
    B, N, P = case['B'], case['N'], case['H'] * case['W']
    obj, cls, masks, scaled, track = out
    stable = torch.from_numpy((g['row_margin'] > 5e-5).all(axis=0))
    d_obj = (obj.reshape(B, N, -1) - torch.from_numpy(g['object_feats']).reshape(B, N, -1)).abs().amax(-1)
    clean = stable & (d_obj < 2e-4)
    flat = masks.reshape(-1)
    idx = torch.from_numpy(g['sample_idx'])
    d = (flat[idx] - torch.from_numpy(g['sample_val'])).abs()
    srow = clean.reshape(-1)[idx // P]
    per_row = torch.zeros(B * N).scatter_reduce(0, idx // P, d, 'amax', include_self=True)
    sampled = torch.zeros(B * N, dtype=torch.bool).index_fill_(0, idx // P, True)
    bits = np.unpackbits(np.packbits(flat.numpy() > 0) ^ g['sign_bits'])[:flat.numel()] & np.unpackbits(g['sign_valid'])[:flat.numel()]
    wrong = torch.from_numpy(bits.astype(bool)).reshape(B, N, P)
    return dict(rows=B * N, stable_share=float(stable.float().mean()), clean_share_of_stable=float(clean.sum()) / float(stable.sum()),
                share_rows_kernels_within_2e4=float((d_obj < 2e-4).float().mean()),
                share_rows_sampled_logits_within_1e3=float((per_row[sampled] < 1e-3).float().mean()),
                worst_clean_kernel_err=float(d_obj[clean].max()), worst_any_kernel_err=float(d_obj.max()),
                worst_clean_sampled_logit_err=float(d[srow].max()) if bool(srow.any()) else 0.0, worst_any_sampled_logit_err=float(d.max()),
                wrong_bits_off_threshold=int(wrong.sum()), wrong_bits_in_clean_rows=int(wrong[clean].sum()))


def cfg2_metrics(base_tr, tr):
    """the quantities of test_cfg2_size_free_running_vs_oracle_flip_budget: a free-running run against the free-running baseline"""
    N = base_tr[0]['new_mask_preds'].shape[1]
    clean = torch.ones(N, dtype=torch.bool)
    rec, total = {}, 0
    for s in range(len(tr)):
        got, ref = tr[s]['new_mask_preds'][0], base_tr[s]['new_mask_preds'][0]
        flip = (got >= THR) != (ref >= THR)
        fc = flip[clean]
        nflip = int(fc.sum())
        total += nflip
        dl = (got - ref).abs().flatten(1).amax(1)
        do = (tr[s]['obj_feat'][0].reshape(N, -1) - base_tr[s]['obj_feat'][0].reshape(N, -1)).abs().amax(1)
        rec.update({f's{s}_clean_rows_in': int(clean.sum()), f's{s}_flipped_bits_clean': nflip, f's{s}_flipped_bits_all': int(flip.sum()),
                    f's{s}_flip_max_dist_to_thr': float((ref[clean][fc] - THR).abs().max()) if nflip else 0.0,
                    f's{s}_worst_clean_logit_err': float(dl[clean].max()), f's{s}_worst_clean_kernel_err': float(do[clean].max()),
                    f's{s}_worst_any_logit_err': float(dl.max())})
        clean = clean & ~flip.flatten(1).any(dim=1)
    rec.update(clean_rows_out=int(clean.sum()), total_flipped_bits_clean=total)
    return rec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--perms', type=int, default=4)
    ap.add_argument('--out', default=os.path.join(ROOT, 'profiles', 'r05_oracle_reorder_noise.json'))
    ap.add_argument('--skip-golden', action='store_true')
    args = ap.parse_args()
    res = dict(_what=__doc__.split('\n\n')[0], _threads_baseline=torch.get_num_threads())
    variants = [('perm %d' % k, dict(perm=torch.randperm(256, generator=torch.Generator().manual_seed(100 + k)))) for k in range(args.perms)]
    base_threads = torch.get_num_threads()
    variants += [('threads %d' % t, dict(threads=t)) for t in (1, 16) if t != base_threads]
    # ---- BASELINE cfg2 size, one frame, S = 3 (the case of test_cfg2_size_free_running_vs_oracle_flip_budget)
    case = dict(C=256, heads=8, ffn=2048, ncls=19, n_thing=2, n_stuff=17, S=3, up=4, nprop=100, N=117, H=128, W=256, B=1, seed=12, video=0)
    _, base_tr = run(case)
    for nm, kw in variants:
        _, tr = run(case, **kw)
        m = cfg2_metrics(base_tr, tr)
        res[f'cfg2_oracle_vs_oracle[{nm}]'] = m
        print(f'cfg2 {nm:10s}: flipped bits (clean rows) per stage {[m[f"s{s}_flipped_bits_clean"] for s in range(3)]}, total {m["total_flipped_bits_clean"]}, '
              f'clean rows at the end {m["clean_rows_out"]} of 117, worst clean-row logit error {max(m[f"s{s}_worst_clean_logit_err"] for s in range(3)):.2e}', flush=True)
    # ---- the cfg5 / cfg4 size goldens (reference outputs): the oracle and its re-ordered runs against the REFERENCE
    if not args.skip_golden:
        for name in ('video_vipseg_big', 'det_ytvis', 'video_vipseg_n216'):
            g, case = load_golden(name)
            for nm, kw in [('baseline', {})] + variants:
                out, _ = run(case, **kw)
                m = golden_metrics(name, out)
                res[f'golden_oracle_vs_reference[{name}][{nm}]'] = m
                print(f'{name} {nm:10s}: rows within 2e-4 {m["share_rows_kernels_within_2e4"]:.4f}, clean share of stable {m["clean_share_of_stable"]:.4f}, '
                      f'wrong bits off threshold {m["wrong_bits_off_threshold"]} (in clean rows {m["wrong_bits_in_clean_rows"]}), '
                      f'worst clean kernel / logit error {m["worst_clean_kernel_err"]:.2e} / {m["worst_clean_sampled_logit_err"]:.2e}', flush=True)
    with open(args.out, 'w') as f:
        json.dump(res, f, indent=1, sort_keys=True)
    print('wrote', args.out)


if __name__ == '__main__':
    main()

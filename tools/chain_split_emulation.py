"""VERDICT r04 item 4, the ACCURACY side (the time side is profiles/r05_chain_two_term_prize.txt): what would the [N x C] chain's results
look like if its GEMMs ran on a TWO-term f16 split (hi + lo, 3 cross products, 4 bytes per operand — what gather / decode use) instead of
the shipped THREE-term bf16 split (6 products, 6 bytes)?  CPU emulation, no GPU: every `F.linear` of the oracle's stage (dynamic / input
layers, gates, fc, attention in/out projections, FFN, fc_cls, fc_mask — exactly the chain; the 1x1 conv, the gather and the decode stay
as they are) is replaced by the split arithmetic with fp32 accumulation:
   fp32      torch's own fp32 GEMM (the reference's arithmetic)
   bf16x3    v = h + m + l (bf16 each), products hh hm mh hl lh mm                       <- shipped (k_gemm_t3, k_chain_*, k_gemm_ks)
   f16x2     v * 2^k = hi + lo (fp16 each; one power of two per tensor puts max|v| at ~2^10), products hh hl lh, result * 2^-(k_a + k_w)
   f16x2-ns  the same without the power-of-two scale (what a kernel without range management would do)
Each against the fp64 oracle on the same inputs (one stage at BASELINE cfg2 size, teacher-forced by construction): max error of the class
logits, the updated kernels and the mask logits, and how many mask bits differ from fp64's off the threshold.
   python tools/chain_split_emulation.py [seeds] [--out profiles/r05_chain_two_term_accuracy.txt]"""
import math
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from helpers import make_case  # noqa: E402
import oracle.knet_oracle as O  # noqa: E402

REAL_LINEAR = F.linear
THR = 8.940696716308594e-08


def pow2(t, target=10):
    m = float(t.abs().max())
    if m == 0.0 or not math.isfinite(m):
        return 1.0
    return 2.0 ** (target - math.frexp(m)[1])


def split(v, dtype, terms):
    out, r = [], v
    for _ in range(terms):
        h = r.to(dtype)
        out.append(h.float())
        r = r - h.float()
    return out


def make_linear(kind):
    def lin(x, w, b=None):
        if x.dtype != torch.float32 or kind == 'fp32':
            return REAL_LINEAR(x, w, b)
        shp = x.shape
        a = x.reshape(-1, shp[-1])
        if kind == 'bf16x3':
            ah, am, al = split(a, torch.bfloat16, 3)
            wh, wm, wl = split(w, torch.bfloat16, 3)
            y = (ah @ wh.t()) + ((ah @ wm.t()) + (am @ wh.t())) + ((ah @ wl.t()) + (al @ wh.t()) + (am @ wm.t()))
        else:
            sa, sw = (pow2(a), pow2(w)) if kind == 'f16x2' else (1.0, 1.0)
            ah, al = split(a * sa, torch.float16, 2)
            wh, wl = split(w * sw, torch.float16, 2)
            y = ((ah @ wh.t()) + ((ah @ wl.t()) + (al @ wh.t()))) * (1.0 / (sa * sw))
        if b is not None:
            y = y + b
        return y.reshape(shp[:-1] + (w.shape[0],))
    return lin


def run(sd, x, pf, mp, cfg, kind):
    tr = []
    F.linear = make_linear(kind)
    try:
        with torch.no_grad():
            O.iter_head_mask_preds(sd, x, pf, mp, cfg, traces=tr)
    finally:
        F.linear = REAL_LINEAR
    return tr[0]


def main():
    seeds = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 3
    out = sys.argv[sys.argv.index('--out') + 1] if '--out' in sys.argv else None
    lines = [__doc__.split('\n   python')[0], '']
    kinds = ('fp32', 'bf16x3', 'f16x2', 'f16x2-ns')
    worst = {k: [0.0, 0.0, 0.0, 0] for k in kinds}
    for seed in range(seeds):
        case = dict(C=256, heads=8, ffn=2048, ncls=19, n_thing=2, n_stuff=17, S=1, up=1, nprop=100, N=117, H=128, W=256, B=1, seed=12 + seed, video=0)
        cfg, sd, x, pf, mp, prev = make_case(case)
        with torch.no_grad():
            tr64 = []
            O.iter_head_mask_preds({k: v.double() for k, v in sd.items()}, x.double(), pf.double(), mp.double(), cfg, traces=tr64)
        ref = tr64[0]
        z64 = ref['new_mask_preds']
        far = (z64 - THR).abs() > 1e-3          # bits whose fp64 logit is not within 1e-3 of the threshold
        for kind in kinds:
            t = run(sd, x, pf, mp, cfg, kind)
            e_cls = float((t['cls_score'].double() - ref['cls_score']).abs().max())
            e_obj = float((t['obj_feat'].double() - ref['obj_feat']).abs().max())
            e_z = float((t['new_mask_preds'].double() - z64).abs().max())
            flips = int((((t['new_mask_preds'] >= THR) != (z64 >= THR)) & far).sum())
            w = worst[kind]
            w[0], w[1], w[2], w[3] = max(w[0], e_cls), max(w[1], e_obj), max(w[2], e_z), w[3] + flips
            line = (f'seed {12 + seed}  {kind:9s}  cls logits {e_cls:.2e}   kernels {e_obj:.2e} (scale {float(ref["obj_feat"].abs().max()):.1f})   '
                    f'mask logits {e_z:.2e} (scale {float(z64.abs().max()):.0f})   mask bits off fp64 (|z - thr| > 1e-3): {flips}')
            print(line, flush=True)
            lines.append(line)
    lines.append('')
    for kind in kinds:
        w = worst[kind]
        lines.append(f'worst over {seeds} seeds  {kind:9s}  cls {w[0]:.2e}   kernels {w[1]:.2e}   mask logits {w[2]:.2e}   far bits flipped {w[3]}')
    lines.append('the parity tests\' teacher-forced bounds at this size: kernels < 1e-4, mask logits < 1e-3 (tests/test_gpu_parity.py)')
    print('\n'.join(lines[-len(kinds) - 1:]))
    if out:
        with open(out, 'w') as f:
            f.write('\n'.join(lines) + '\n')


if __name__ == '__main__':
    main()

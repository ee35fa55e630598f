"""Error of every form of the [N x C] chain against the torch fp64 oracle, one stage at BASELINE cfg2 width (C = 256, ff = 2048): the
few-row chain (vkn_ksplit.hip), one launch per GEMM (k_gemm_t3), the persistent row owners (k_chain_*), the exact-fp32 MFMA chain, and
the torch fp32 oracle itself.  All forms must sit at the same distance from fp64 (they differ from each other only by summation order).
   python tools/chain_accuracy.py [seeds]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import vkn_import  # noqa: E402

vkn = vkn_import.load()
from helpers import make_case  # noqa: E402
from test_host_logic import _cfg  # noqa: E402
import oracle.knet_oracle as O  # noqa: E402

DEV = 'cuda:0'


def main():
    seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    C, heads, H, W, N, ff, ncls = 256, 8, 16, 32, 117, 2048, 19
    forms = (('few-row', vkn.ops.FLAG_CHAIN_KSPLIT), ('launch-per-GEMM', vkn.ops.FLAG_CHAIN_LAUNCHES), ('persistent', vkn.ops.FLAG_CHAIN_PERSISTENT),
             ('exact-fp32 GEMMs', vkn.ops.FLAG_EXACT_GEMM))
    print('max |error| against the fp64 oracle (one stage, teacher-forced inputs): cls logits | updated kernels | mask logits (scale)')
    for seed in range(seeds):
        for B in (1, 4):
            kw = dict(C=C, heads=heads, ffn=ff, ncls=ncls, n_thing=2, n_stuff=17, S=1, up=1, nprop=N - 17)
            case = dict(kw, N=N, H=H, W=W, B=B, seed=900 + seed, video=0)
            head = vkn.build_head(_cfg(False, **kw))
            cfg, sd, x, pf, mp, prev = make_case(case)
            head.load_state_dict(sd, strict=True)
            head = head.to(DEV).eval()
            with torch.no_grad():
                tr64, tr32 = [], []
                O.iter_head_mask_preds({k: v.double() for k, v in sd.items()}, x.double(), pf.double(), mp.double(), cfg, traces=tr64)
                O.iter_head_mask_preds(sd, x, pf, mp, cfg, traces=tr32)
            ref = tr64[0]
            dims = head.mask_head[0].make_dims(B, N, H, W)
            pack = head.mask_head[0].stage_pack(torch.device(DEV))
            args = (dims, pack, x.to(DEV), pf.reshape(B, N, C).to(DEV), mp.to(DEV))

            def err(cls, masks, obj):
                return (float((cls.double().cpu() - ref['cls_score']).abs().max()), float((obj.double().cpu().reshape(B, N, C) - ref['obj_feat'].reshape(B, N, C)).abs().max()),
                        float((masks.double().cpu() - ref['new_mask_preds']).abs().max()))
            sc = float(ref['new_mask_preds'].abs().max())
            e = err(tr32[0]['cls_score'], tr32[0]['new_mask_preds'], tr32[0]['obj_feat'])
            print(f'seed {seed} B={B}  {"torch fp32 oracle":18s} {e[0]:.2e} | {e[1]:.2e} | {e[2]:.2e} ({sc:.1f})')
            for nm, fl in forms:
                cls, masks, obj, _, _ = vkn.ops.stage_forward(*args, flags=fl)
                e = err(cls, masks, obj)
                print(f'seed {seed} B={B}  {nm:18s} {e[0]:.2e} | {e[1]:.2e} | {e[2]:.2e}')


if __name__ == '__main__':
    main()

"""Round-5 measurements of the few-row regime: the three forms of the [N x C] chain (few-row column-spread phases = vkn_ksplit.hip,
one launch per GEMM = k_gemm_t3, persistent row owners = k_chain_*) — chain alone (`vkn_stage_chain_f32`) and whole head steps at
1 / 2 / 4 / 8 / 16 / 32 frames per call.   python tools/perf_r05.py [--what chain,head] [--frames 1,2,4,8]"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import vkn_import  # noqa: E402

vkn = vkn_import.load()
DEV = 'cuda:0'
if '--debug-lib' in sys.argv:
    sys.argv.remove('--debug-lib')
    vkn._lib.build_debug()
    vkn._lib.use_debug()


def timeit(fn, iters=50, warm=10):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3   # us


FORMS = (('few-row', 'FLAG_CHAIN_KSPLIT'), ('launch-per-GEMM', 'FLAG_CHAIN_LAUNCHES'), ('persistent', 'FLAG_CHAIN_PERSISTENT'))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--what', default='chain,head')
    ap.add_argument('--frames', default='1,2,4,8,16,32')
    ap.add_argument('--reps', type=int, default=2)
    args = ap.parse_args()
    what = args.what.split(',')
    frames = [int(f) for f in args.frames.split(',')]
    N, C, H, W = 117, 256, 128, 256
    cfg = vkn.configs.roi_head_cfg(True, C=C, heads=8, ffn=2048, ncls=19, n_thing=2, n_stuff=17, S=3, up=4, nprop=100)
    head = vkn.build_head(cfg)
    torch.manual_seed(0)
    head.init_weights()
    head = head.to(DEV).eval()
    g = torch.Generator(device='cpu').manual_seed(1)
    if 'chain' in what:
        print('== chain alone (vkn_stage_chain_f32), us per stage ==')
        for B in frames:
            dims = head.mask_head[0].make_dims(B, N, H, W)
            pack = head.mask_head[0].stage_pack(torch.device(DEV))
            xf = (torch.randn(B, N, C, generator=g) * 50).to(DEV)
            ob = torch.randn(B, N, C, generator=g).to(DEV)
            for rep in range(args.reps):
                ts = [timeit(lambda: vkn.ops.stage_chain(dims, pack, xf, ob, flags=getattr(vkn.ops, fl))) for _, fl in FORMS]
                print(f'B={B:3d} rows={B * N:5d}  ' + '   '.join(f'{nm} {t:7.1f}' for (nm, _), t in zip(FORMS, ts)))
    if 'wgcap' in what:     # needs --debug-lib: the few-row chain's workgroup-count cap (tile shape heuristic), chain alone
        print('== few-row chain alone by VKN_KS_WGCAP (workgroups per launch above which tiles get fatter), us per stage ==')
        for B in frames:
            dims = head.mask_head[0].make_dims(B, N, H, W)
            pack = head.mask_head[0].stage_pack(torch.device(DEV))
            xf = (torch.randn(B, N, C, generator=g) * 50).to(DEV)
            ob = torch.randn(B, N, C, generator=g).to(DEV)
            row = []
            for cap in (256, 192, 128, 96, 256):
                os.environ['VKN_KS_WGCAP'] = str(cap)
                t = timeit(lambda: vkn.ops.stage_chain(dims, pack, xf, ob, flags=vkn.ops.FLAG_CHAIN_KSPLIT))
                row.append(f'cap {cap}: {t:6.1f}')
            t3 = timeit(lambda: vkn.ops.stage_chain(dims, pack, xf, ob, flags=vkn.ops.FLAG_CHAIN_LAUNCHES))
            print(f'B={B:3d} rows={B * N:5d}  ' + '   '.join(row) + f'   launch-per-GEMM {t3:6.1f}')
        os.environ.pop('VKN_KS_WGCAP', None)
    if 'abl11' in what:      # needs --debug-lib.  VERDICT r04 item 4, the prize measured before the work: the persistent chain with TWO
        # split terms per operand (4 bytes / weight, 3 products per operand pair instead of 6 bytes / 6 products) — VKN_CHAIN_ABL=11 keeps
        # everything else (activation images, epilogues, ring) and produces WRONG numbers (bf16 x 2 precision); only its time counts
        print('== persistent chain, three-term split (shipped) vs the traffic / MFMA count of a two-term split (VKN_CHAIN_ABL=11) ==')
        for B in (32, 64):
            dims = head.mask_head[0].make_dims(B, N, H, W)
            pack = head.mask_head[0].stage_pack(torch.device(DEV))
            xf = (torch.randn(B, N, C, generator=g) * 50).to(DEV)
            ob = torch.randn(B, N, C, generator=g).to(DEV)
            x = torch.randn(B, C, H, W, generator=g).to(DEV) if B == 32 else None
            pf = torch.randn(B, N, C, 1, 1, generator=g).to(DEV) if B == 32 else None
            mp = (torch.randn(B, N, H, W, generator=g) * 4).to(DEV) if B == 32 else None
            prev = torch.randn(B, N, C, 1, 1, generator=g).to(DEV) if B == 32 else None
            for rep in range(2):
                for abl in (0, 11):
                    os.environ['VKN_CHAIN_ABL'] = str(abl)
                    t = timeit(lambda: vkn.ops.stage_chain(dims, pack, xf, ob, flags=vkn.ops.FLAG_CHAIN_PERSISTENT | vkn.ops.FLAG_CHAIN_BF16X3))
                    line = f'B={B:3d} rows={B * N:5d} ABL={abl:2d}  chain alone {t:7.1f} us per stage'
                    if x is not None:
                        with torch.no_grad():
                            th = timeit(lambda: head._head_forward(x, pf, mp, prev, want_track=True, flags=vkn.ops.FLAG_CHAIN_BF16X3), iters=20, warm=5)
                        line += f'   head step {th / 1e3:7.3f} ms = {B / th * 1e6:7.0f} frames/s'
                    print(line)
        os.environ['VKN_CHAIN_ABL'] = '0'
    if 'h2' in what:       # VERDICT r04 item 4 built: the persistent chain on the two-term fp16 split (vkn_chain_h2.hip) against the bf16 x 3 form
        print('== persistent chain: three-term bf16 split (6 B / weight, 6 MFMAs) vs two-term fp16 split (the default since round 5: 4 B, 3 MFMAs) ==')
        for B in (32, 64):
            dims = head.mask_head[0].make_dims(B, N, H, W)
            pack = head.mask_head[0].stage_pack(torch.device(DEV))
            xf = (torch.randn(B, N, C, generator=g) * 50).to(DEV)
            ob = torch.randn(B, N, C, generator=g).to(DEV)
            x = torch.randn(B, C, H, W, generator=g).to(DEV) if B == 32 else None
            pf = torch.randn(B, N, C, 1, 1, generator=g).to(DEV) if B == 32 else None
            mp = (torch.randn(B, N, H, W, generator=g) * 4).to(DEV) if B == 32 else None
            prev = torch.randn(B, N, C, 1, 1, generator=g).to(DEV) if B == 32 else None
            for rep in range(args.reps):
                for nm, fl in (('bf16x3', vkn.ops.FLAG_CHAIN_PERSISTENT | vkn.ops.FLAG_CHAIN_BF16X3), ('fp16x2', vkn.ops.FLAG_CHAIN_PERSISTENT)):
                    t = timeit(lambda: vkn.ops.stage_chain(dims, pack, xf, ob, flags=fl))
                    line = f'B={B:3d} rows={B * N:5d} {nm}  chain alone {t:7.1f} us per stage'
                    if x is not None:
                        with torch.no_grad():
                            th = timeit(lambda: head._head_forward(x, pf, mp, prev, want_track=True, flags=fl & vkn.ops.FLAG_CHAIN_BF16X3), iters=20, warm=5)
                        line += f'   head step (default policy) {th / 1e3:7.3f} ms = {B / th * 1e6:7.0f} frames/s'
                    print(line)
    if 'join' in what:     # where the side-stream link joins the main stream: behind the upsample (default) vs before it (VKN_FLAG_JOIN_EARLY)
        print('== link join behind (default) vs before (VKN_FLAG_JOIN_EARLY) the x4 upsample, ms per call ==')
        for B in frames:
            x = torch.randn(B, C, H, W, generator=g).to(DEV)
            pf = torch.randn(B, N, C, 1, 1, generator=g).to(DEV)
            mp = (torch.randn(B, N, H, W, generator=g) * 4).to(DEV)
            prev = torch.randn(B, N, C, 1, 1, generator=g).to(DEV)
            for rep in range(args.reps):
                row = []
                for nm, fl in (('join behind upsample', 0), ('join before upsample', vkn.ops.FLAG_JOIN_EARLY)):
                    with torch.no_grad():
                        t = timeit(lambda: head._head_forward(x, pf, mp, prev, want_track=True, flags=fl), iters=40, warm=8)
                    row.append(f'{nm} {t / 1e3:7.3f} ms = {B / t * 1e6:7.0f} f/s')
                print(f'B={B:3d}  ' + '   '.join(row))
    if 'head' in what:
        print('== whole head step (3 stages + link + x4 upsample), ms per call ==')
        for B in frames:
            x = torch.randn(B, C, H, W, generator=g).to(DEV)
            pf = torch.randn(B, N, C, 1, 1, generator=g).to(DEV)
            mp = (torch.randn(B, N, H, W, generator=g) * 4).to(DEV)
            prev = torch.randn(B, N, C, 1, 1, generator=g).to(DEV)
            for rep in range(args.reps):
                row = []
                for nm, fl in (('default', None),) + FORMS:
                    with torch.no_grad():
                        t = timeit(lambda: head._head_forward(x, pf, mp, prev, want_track=True, flags=(getattr(vkn.ops, fl) if fl else 0)),
                                   iters=30, warm=5)
                    row.append(f'{nm} {t / 1e3:7.3f} ms = {B / t * 1e6:7.0f} f/s')
                print(f'B={B:3d}  ' + '   '.join(row))


if __name__ == '__main__':
    main()

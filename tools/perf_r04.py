"""Round-4 measurements of the [N x C] chain: the persistent row-owner kernels (k_chain_a + attention + k_chain_c, default) against
the launch-per-GEMM chain (VKN_FLAG_CHAIN_LAUNCHES), chain alone (`vkn_stage_chain_f32`: x_feat in, decode kernels out) and whole
head steps, at several frames per call.   python tools/perf_r04.py [--what chain|head|all]"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import vkn_import  # noqa: E402

vkn = vkn_import.load()
DEV = 'cuda:0'
if '--debug-lib' in sys.argv:       # time-attribution variants (VKN_CHAIN_ABL ...) exist only in lib/libvkn_debug.so
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--what', default='all')
    args = ap.parse_args()
    N, C, H, W = 117, 256, 128, 256
    cfg = vkn.configs.roi_head_cfg(True, C=C, heads=8, ffn=2048, ncls=19, n_thing=2, n_stuff=17, S=3, up=4, nprop=100)
    head = vkn.build_head(cfg)
    torch.manual_seed(0)
    head.init_weights()
    head = head.to(DEV).eval()
    g = torch.Generator(device='cpu').manual_seed(1)
    if args.what in ('chain', 'all'):
        print('== chain alone (vkn_stage_chain_f32), us per stage ==')
        for B in (1, 2, 4, 8, 16, 32, 64):
            dims = head.mask_head[0].make_dims(B, N, H, W)
            pack = head.mask_head[0].stage_pack(torch.device(DEV))
            xf = (torch.randn(B, N, C, generator=g) * 50).to(DEV)
            ob = torch.randn(B, N, C, generator=g).to(DEV)
            t_new = timeit(lambda: vkn.ops.stage_chain(dims, pack, xf, ob))
            t_old = timeit(lambda: vkn.ops.stage_chain(dims, pack, xf, ob, flags=vkn.ops.FLAG_CHAIN_LAUNCHES))
            print(f'B={B:3d} rows={B * N:5d}  persistent {t_new:8.1f}   launch-per-GEMM {t_old:8.1f}   ratio {t_old / t_new:5.2f}')
    if args.what == 'abl':           # needs --debug-lib
        print('== chain alone, time attribution (VKN_CHAIN_ABL: 1 no MFMA, 2 no weight loads, 3 no fragment reads, 4 neither, 5|6|7 no MFMA + ring 2 | ring 8 | nt loads), us per stage ==')
        for B in (1, 32):
            dims = head.mask_head[0].make_dims(B, N, H, W)
            pack = head.mask_head[0].stage_pack(torch.device(DEV))
            xf = (torch.randn(B, N, C, generator=g) * 50).to(DEV)
            ob = torch.randn(B, N, C, generator=g).to(DEV)
            for abl in (0, 1, 2, 4, 8, 9, 10, 0):
                os.environ['VKN_CHAIN_ABL'] = str(abl)
                t = timeit(lambda: vkn.ops.stage_chain(dims, pack, xf, ob))
                print(f'B={B:3d} ABL={abl}  {t:8.1f}')
        os.environ['VKN_CHAIN_ABL'] = '0'
    if args.what == 'decpol':        # needs --debug-lib: cache policy of the decode's x loads / output stores (VKN_DECODE_ABL 4..9)
        B = 32
        x = torch.randn(B, C, H, W, generator=g).to(DEV)
        pf = torch.randn(B, N, C, 1, 1, generator=g).to(DEV)
        mp = (torch.randn(B, N, H, W, generator=g) * 4).to(DEV)
        prev = torch.randn(B, N, C, 1, 1, generator=g).to(DEV)
        kern = torch.randn(B, N, C, generator=g).to(DEV)
        hi, lo = vkn.ops.split_planes(kern)
        kb = torch.randn(B, N, generator=g).to(DEV)
        out = torch.empty(B, N, H, W, device=DEV)
        alg = B * H * W * (C * 4 + N * 4)
        names = {0: 'sc0 nt (shipped)', 4: 'plain', 6: 'nt', 7: 'sc1', 8: 'sc1 nt', 9: 'sc0 sc1 nt', 5: 'shipped loads, nt stores'}
        for rep in range(2):
            for abl in (0, 4, 6, 7, 8, 9, 5, 0):
                os.environ['VKN_DECODE_ABL'] = str(abl)
                t = timeit(lambda: vkn.ops.mask_decode_planes(x, hi, lo, N, kb, out), iters=40, warm=10)
                with torch.no_grad():
                    th = timeit(lambda: head._head_forward(x, pf, mp, prev, want_track=True), iters=15, warm=4)
                print(f'decode x-load policy {names[abl]:26s} isolated loop {t:7.1f} us = {alg / t / 1e6 / 8:.3f} of 8 TB/s   head step {th / 1e3:7.3f} ms')
        os.environ['VKN_DECODE_ABL'] = '0'
    if args.what == 'upsample':      # needs --debug-lib: x4 upsample, taps from global (shipped, VKN_UPSAMPLE=14) vs input rows through LDS (54)
        B = 32
        z = (torch.randn(B, N, H, W, generator=g) * 4).to(DEV)
        outs = {}
        for rep in range(3):
            for mode in (14, 54, 74, 34):
                os.environ['VKN_UPSAMPLE'] = str(mode)
                t = timeit(lambda: vkn.ops.upsample_bilinear(z, 4), iters=20, warm=5)
                outs[mode] = vkn.ops.upsample_bilinear(z, 4)
                nm = {14: 'taps from global (shipped)', 54: 'input rows through LDS', 74: 'all rows requested up front', 34: 'write-only ablation'}[mode]
                print(f'upsample x4, {nm:30s} {t:8.1f} us = {B * N * H * W * 16 * 4 / t / 1e6:.2f} TB/s of writes')
        print('bit-identical to the shipped kernel:', bool(torch.equal(outs[14], outs[54])))
        os.environ['VKN_UPSAMPLE'] = '14'
    if args.what == 'fused':         # needs --debug-lib: the fused decode -> gather pass, packed f16 split (shipped) vs the round-3 split (VKN_FUSED=19)
        B = 32
        x = torch.randn(B, C, H, W, generator=g).to(DEV)
        kern = torch.randn(B, N, C, generator=g).to(DEV)
        hi, lo = vkn.ops.split_planes(kern)
        kb = torch.randn(B, N, generator=g).to(DEV)
        for rep in range(3):
            for var, nm in ((10, 'vkn_split_f16x2 (shipped)'), (19, 'two vkn_split_f16 per pixel pair (round 3)')):
                os.environ['VKN_FUSED'] = str(var)
                t = timeit(lambda: vkn.ops.decode_gather(x, hi, lo, N, kb), iters=40, warm=10)
                print(f'fused pass, {nm:44s} {t:7.1f} us = {B * C * H * W * 4 / t / 1e6 / 8:.3f} of 8 TB/s by x bytes')
        os.environ['VKN_FUSED'] = '10'
    if args.what in ('head', 'all'):
        print('== whole head step (3 stages + link + x4 upsample), ms per call ==')
        for B in (1, 8, 32):
            x = torch.randn(B, C, H, W, generator=g).to(DEV)
            pf = torch.randn(B, N, C, 1, 1, generator=g).to(DEV)
            mp = (torch.randn(B, N, H, W, generator=g) * 4).to(DEV)
            prev = torch.randn(B, N, C, 1, 1, generator=g).to(DEV)
            for nm, fl in (('persistent', 0), ('launch-per-GEMM', vkn.ops.FLAG_CHAIN_LAUNCHES)):
                with torch.no_grad():
                    t = timeit(lambda: head._head_forward(x, pf, mp, prev, want_track=True, flags=fl), iters=30, warm=5)
                print(f'B={B:3d} {nm:16s} {t / 1e3:8.3f} ms  -> {B / t * 1e6:8.1f} frames/s')


if __name__ == '__main__':
    main()

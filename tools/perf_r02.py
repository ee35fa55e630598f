#!/usr/bin/env python3
"""Round-2 kernel A/B on one MI355X (debug library: reads VKN_* knobs).  Prints one line per measurement.

    python tools/perf_r02.py [--frames 8,32] [--what decode,fused,head,upsample]

decode:   k_decode_mfma (8-byte accesses, VKN_DECODE4=0) vs k_decode4 (16-byte accesses), logits output, cfg2 shape
fused:    stages' hand-off through the fused decode->gather pass vs bit words (flag 16) vs logits (flag 4): whole head, no upsample
head:     whole bench step (S = 3 + link + x4 upsample) at B = 1 / 8 / 32
upsample: x4 bilinear upsample alone
"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def timeit(fn, reps=20, warm=5):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3   # us


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--frames', default='8,32')
    ap.add_argument('--what', default='decode,fused,head,upsample')
    ap.add_argument('--release', action='store_true', help='use the release library (no env knobs)')
    ap.add_argument('--xdtype', default='fp32', choices=['fp32', 'fp16', 'bf16'], help='storage type of the feature map x')
    args = ap.parse_args()
    import vkn_import
    vkn = vkn_import.load()
    if not args.release:
        vkn._lib.build_debug()
        vkn._lib.use_debug()
    import bench
    dev = torch.device('cuda', 0)
    what = args.what.split(',')
    N, C, H, W = 117, 256, 128, 256
    P = H * W
    head = bench.build_head(vkn, dev)
    for B in [int(v) for v in args.frames.split(',')]:
        x, pf, mp = bench.synth_inputs(B, dev, 0)
        x = x.to({'fp32': torch.float32, 'fp16': torch.float16, 'bf16': torch.bfloat16}[args.xdtype])
        xbytes = x.element_size()
        alg = B * P * (C * xbytes + N * 4)
        if 'decode' in what:
            kern = torch.randn(B, N, C, device=dev)
            hi, lo = vkn.ops.split_planes(kern)
            kb = torch.randn(B, N, device=dev)
            out = torch.empty(B, N, H, W, device=dev)
            res = {}
            os.environ['VKN_DECODE4'] = '0'
            for opt in (0, 1, 0):
                os.environ['VKN_DECODE_OPT'] = str(opt)
                t = timeit(lambda: vkn.ops.mask_decode_planes(x, hi, lo, N, kb, out), reps=40)
                res[opt] = out.clone()
                print(f'decode B={B} k_decode_mfma OPT={opt}: {t:8.1f} us  {alg / t / 1e6:7.3f} TB/s  frac {alg / t / 1e6 / 8:.3f}'
                      f'  bit-identical to OPT=0: {torch.equal(res[0], res[opt])}', flush=True)
            os.environ['VKN_DECODE_OPT'] = '1'
            for ppw in (1024, 2048, 4096, 8192, 512):
                os.environ['VKN_DECODE_PXWG'] = str(ppw)
                t = timeit(lambda: vkn.ops.mask_decode_planes(x, hi, lo, N, kb, out), reps=40)
                print(f'decode B={B} k_decode_mfma OPT=1 px/wg={ppw}: {t:8.1f} us  {alg / t / 1e6:7.3f} TB/s  frac {alg / t / 1e6 / 8:.3f}', flush=True)
            os.environ.pop('VKN_DECODE_PXWG')
            os.environ['VKN_DECODE_OPT'] = '0'
            os.environ['VKN_DECODE4'] = '1'
            t = timeit(lambda: vkn.ops.mask_decode_planes(x, hi, lo, N, kb, out), reps=40)
            print(f'decode B={B} k_decode4: {t:8.1f} us  {alg / t / 1e6:7.3f} TB/s  bit-identical: {torch.equal(res[0], out)}', flush=True)
            os.environ['VKN_DECODE4'] = '0'
            # the fused decode -> gather pass alone: one-wave-per-SIMD (VKN_FUSED8=0) vs two (1); bytes = x only
            xb = B * P * C * xbytes
            ref = None
            for f8 in ('0', '1', '2', '2'):
                os.environ['VKN_FUSED'] = f8
                t = timeit(lambda: vkn.ops.decode_gather(x, hi, lo, N, kb), reps=20)
                r = vkn.ops.decode_gather(x, hi, lo, N, kb)
                same = True if ref is None else (torch.equal(r[0], ref[0]) and torch.equal(r[1], ref[1]))
                ref = r if ref is None else ref
                print(f'fused decode->gather B={B} variant={f8} (0 dg, 1 dg8, 2 dgs): {t:8.1f} us  {xb / t / 1e6:7.3f} TB/s of x  same: {same}', flush=True)
            os.environ['VKN_FUSED'] = '2'
            t = timeit(lambda: vkn.ops.mask_gather(x, mp), reps=30)
            print(f'gather(logits)+reduce B={B}: {t:8.1f} us  {alg / t / 1e6:7.3f} TB/s', flush=True)
            del out, res
        last = head.mask_head[-1]
        dims = last.make_dims(B, N, H, W)
        packs = [h.stage_pack(dev) for h in head.mask_head]
        pfr = pf.reshape(B, N, C)
        if 'fused' in what:
            outs = {}
            for name, fl in (('fused', 0), ('bits', 16), ('logits', 4)):
                t = timeit(lambda: vkn.ops.head_forward(dims, packs, x, pfr, mp, None, 1, flags=fl), reps=10, warm=3)
                outs[name] = vkn.ops.head_forward(dims, packs, x, pfr, mp, None, 1, flags=fl)
                print(f'head(no upsample, no link) B={B} handoff={name}: {t:8.1f} us  {B / t * 1e6:9.1f} frames/s', flush=True)
            same = all(torch.equal(a, b) for a, b in zip(outs['fused'][:3], outs['bits'][:3]))
            same2 = all(torch.equal(a, b) for a, b in zip(outs['fused'][:3], outs['logits'][:3]))
            print(f'head B={B}: fused == bits: {same}; fused == logits: {same2}', flush=True)
            del outs
        if 'head' in what:
            fp = torch.zeros(1, N, C, device=dev)
            t = timeit(lambda: vkn.ops.head_forward(dims, packs, x, pfr, mp, None, 4, clip_first_prev=fp), reps=10, warm=3)
            print(f'bench step (S=3 + link + x4) B={B}: {t:8.1f} us  {B / t * 1e6:9.1f} frames/s', flush=True)
            t = timeit(lambda: vkn.ops.head_forward(dims, packs, x, pfr, mp, None, 4, clip_first_prev=fp, flags=32), reps=10, warm=3)
            print(f'bench step, link on the caller stream (flag 32) B={B}: {t:8.1f} us  {B / t * 1e6:9.1f} frames/s', flush=True)
            t = timeit(lambda: vkn.ops.head_forward(dims, packs, x, pfr, mp, None, 1, clip_first_prev=fp), reps=10, warm=3)
            print(f'bench step without x4 upsample B={B}: {t:8.1f} us  {B / t * 1e6:9.1f} frames/s', flush=True)
        if 'gemmabl' in what and not args.release:
            fp = torch.zeros(1, N, C, device=dev)
            for abl, nm in ((0, 'real'), (1, 'no K loop'), (2, 'no row epilogue'), (3, 'no MFMAs (k_gemm_r3)'), (4, 'no A path in the loop (r3)'), (5, 'no barriers in the loop (r3)'), (6, 'no weight loads in the loop (r3)'), (0, 'real')):
                os.environ['VKN_GEMM_ABL'] = str(abl)
                t = timeit(lambda: vkn.ops.head_forward(dims, packs, x, pfr, mp, None, 1, clip_first_prev=fp), reps=10, warm=3)
                print(f'head without upsample B={B}, k_gemm_s3 ablation {abl} ({nm}): {t:8.1f} us  (24 k_gemm_s3 launches per step)', flush=True)
            os.environ.pop('VKN_GEMM_ABL')
        if 'gemmx16' in what and not args.release:
            fp = torch.zeros(1, N, C, device=dev)
            one = [packs[0]]
            outs = {}
            for x16 in (0, 1, 0, 1):
                os.environ['VKN_GEMM_X16'] = str(x16)
                t = timeit(lambda: vkn.ops.head_forward(dims, packs, x, pfr, mp, None, 1, clip_first_prev=fp), reps=10, warm=3)
                outs[x16] = vkn.ops.head_forward(dims, one, x, pfr, mp, None, 1)
                print(f'head without upsample B={B}, GEMMs {"k_gemm_x16 (16 waves, intra-WG split-K)" if x16 else "k_gemm_s3"}: {t:8.1f} us', flush=True)
            d = [float((a - b).abs().max()) for a, b in zip(outs[0][:3], outs[1][:3])]
            print(f'  one stage, max |diff| of (kernels, cls, mask logits) between the two: {d}', flush=True)
            os.environ.pop('VKN_GEMM_X16')
        if 'gemmr3' in what and not args.release:
            fp = torch.zeros(1, N, C, device=dev)
            outs = {}
            for r3 in (0, 1, 0, 1):
                os.environ['VKN_GEMM_R3'] = str(r3)
                t = timeit(lambda: vkn.ops.head_forward(dims, packs, x, pfr, mp, None, 1, clip_first_prev=fp), reps=10, warm=3)
                outs[r3] = vkn.ops.head_forward(dims, packs, x, pfr, mp, None, 1, clip_first_prev=fp)
                print(f'head without upsample B={B}, GEMM weights {"global -> registers (k_gemm_r3)" if r3 else "global -> LDS -> registers (k_gemm_s3)"}: {t:8.1f} us', flush=True)
            same = all((a is None and b is None) or torch.equal(a, b) for a, b in zip(outs[0], outs[1]))
            print(f'  every output bit-identical: {same}', flush=True)
            os.environ.pop('VKN_GEMM_R3')
        if 'ffnabl' in what and not args.release:
            fp = torch.zeros(1, N, C, device=dev)
            for abl in (0, 1, 2, 0):
                os.environ['VKN_FFN_ABL'] = str(abl)
                t = timeit(lambda: vkn.ops.head_forward(dims, packs, x, pfr, mp, None, 1, clip_first_prev=fp), reps=10, warm=3)
                print(f'head without upsample B={B}, k_ffn_fused ablation {abl} (1 = GEMM 2 without waits, 2 = no GEMM 2): {t:8.1f} us  (4 launches per step)', flush=True)
            os.environ.pop('VKN_FFN_ABL')
        if 'ffnhs' in what and not args.release:
            fp = torch.zeros(1, N, C, device=dev)
            for hs in (0, 1, 2, 4, 0):
                os.environ['VKN_FFN_HS'] = str(hs)
                t = timeit(lambda: vkn.ops.head_forward(dims, packs, x, pfr, mp, None, 1, clip_first_prev=fp), reps=10, warm=3)
                print(f'head without upsample B={B}, FFN hidden split {hs} (0 = policy): {t:8.1f} us', flush=True)
            os.environ.pop('VKN_FFN_HS')
        if 'lastchunk' in what and not args.release:
            fp = torch.zeros(1, N, C, device=dev)
            for ch in (0, 16, 8, 4, 0):
                os.environ['VKN_LAST_CHUNK'] = str(ch)
                t = timeit(lambda: vkn.ops.head_forward(dims, packs, x, pfr, mp, None, 4, clip_first_prev=fp), reps=10, warm=3)
                print(f'bench step B={B}, last decode + upsample in chunks of {ch} frames (0 = whole batch): {t:8.1f} us', flush=True)
            os.environ.pop('VKN_LAST_CHUNK')
        if 'upsample' in what:
            m = torch.randn(B, N, H, W, device=dev)
            t = timeit(lambda: vkn.ops.upsample_bilinear(m, 4), reps=10, warm=3)
            wb = B * N * P * 16 * 4
            print(f'upsample x4 B={B}: {t:8.1f} us  {wb / t / 1e6:7.3f} TB/s of writes', flush=True)
            if not args.release:
                ref = vkn.ops.upsample_bilinear(m, 4)
                for mode in tuple(int(v) for v in os.environ.get('UPMODES', '24,11,64,74,34').split(',')):   # 1RX: k_upsample_f (fill pattern), R code 0/1/2 = 2/4/8 rows, X = 2*xmap + nt;  plain stores, one row group per WG, nontemporal input loads, all loads up front, write-only   # 2x plain stores, x1 one row group, x8 / x9 = 8 / 32 groups per WG, 3x write-only
                    os.environ['VKN_UPSAMPLE'] = str(mode)
                    t = timeit(lambda: vkn.ops.upsample_bilinear(m, 4), reps=10, warm=3)
                    same = bool(torch.equal(vkn.ops.upsample_bilinear(m, 4), ref)) if mode >= 100 else None
                    print(f'upsample x4 B={B} mode={mode}: {t:8.1f} us  {wb / t / 1e6:7.3f} TB/s of writes' + ('' if same is None else f'  bit-identical to the shipped kernel: {same}'), flush=True)
                os.environ.pop('VKN_UPSAMPLE')
            o = torch.empty(B, N, H * 4, W * 4, device=dev)
            t = timeit(lambda: o.fill_(1.0), reps=10, warm=3)
            print(f'torch fill_ of the same bytes B={B}: {t:8.1f} us  {wb / t / 1e6:7.3f} TB/s', flush=True)
            del o, m
        del x, pf, mp
        torch.cuda.empty_cache()


if __name__ == '__main__':
    main()

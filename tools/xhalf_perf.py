#!/usr/bin/env python3
"""x-streaming kernels and the whole bench step with x stored as fp32 / fp16 / bf16 (release library), one MI355X.
    python tools/xhalf_perf.py [--frames 32]"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools'))
from perf_r02 import timeit  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--frames', default='1,8,32')
    args = ap.parse_args()
    import vkn_import
    vkn = vkn_import.load()
    import bench
    dev = torch.device('cuda', 0)
    N, C, H, W = 117, 256, 128, 256
    P = H * W
    head = bench.build_head(vkn, dev)
    for B in [int(v) for v in args.frames.split(',')]:
        x32, pf, mp = bench.synth_inputs(B, dev, 0)
        kern = torch.randn(B, N, C, device=dev)
        hi, lo = vkn.ops.split_planes(kern)
        kb = torch.randn(B, N, device=dev)
        out = torch.empty(B, N, H, W, device=dev)
        dims = head.mask_head[-1].make_dims(B, N, H, W)
        packs = [h.stage_pack(dev) for h in head.mask_head]
        pfr = pf.reshape(B, N, C)
        fp = torch.zeros(1, N, C, device=dev)
        for name, dt in (('fp32', torch.float32), ('fp16', torch.float16), ('bf16', torch.bfloat16)):
            x = x32.to(dt)
            eb = x.element_size()
            alg = B * P * (C * eb + N * 4)
            t = timeit(lambda: vkn.ops.mask_decode_planes(x, hi, lo, N, kb, out), reps=30)
            print(f'B={B:2d} x={name}: decode        {t:8.1f} us  {alg / t / 1e6:6.3f} TB/s  frac of 8 TB/s {alg / t / 8e6:.3f}', flush=True)
            t = timeit(lambda: vkn.ops.mask_gather(x, mp), reps=30)
            print(f'B={B:2d} x={name}: gather+reduce {t:8.1f} us  {alg / t / 1e6:6.3f} TB/s', flush=True)
            t = timeit(lambda: vkn.ops.decode_gather(x, hi, lo, N, kb), reps=20)
            print(f'B={B:2d} x={name}: fused dec->gat {t:7.1f} us  {B * P * C * eb / t / 1e6:6.3f} TB/s of x', flush=True)
            t = timeit(lambda: vkn.ops.head_forward(dims, packs, x, pfr, mp, None, 4, clip_first_prev=fp), reps=10, warm=3)
            print(f'B={B:2d} x={name}: bench step (S=3 + link + x4) {t:8.1f} us  {B / t * 1e6:9.1f} frames/s', flush=True)
            del x
        del x32, pf, mp, out
        torch.cuda.empty_cache()


if __name__ == '__main__':
    main()

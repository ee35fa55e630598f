#!/usr/bin/env python3
"""Consecutive clip steps in flight on K HIP streams (step i on stream i % K): does the latency-bound update chain of one step hide
under the HBM-bound upsample of another?   python tools/inflight_probe.py [--frames 32]"""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--frames', type=int, default=32)
    ap.add_argument('--steps', type=int, default=60)
    ap.add_argument('--debug', action='store_true', help='debug library (reads VKN_* knobs, e.g. VKN_GEMM_R3=1)')
    args = ap.parse_args()
    import vkn_import
    vkn = vkn_import.load()
    if args.debug:
        vkn._lib.build_debug()
        vkn._lib.use_debug()
    import bench
    dev = torch.device('cuda', 0)
    N, C, H, W = 117, 256, 128, 256
    B = args.frames
    head = bench.build_head(vkn, dev)
    x, pf, mp = bench.synth_inputs(B, dev, 0)
    pfr = pf.reshape(B, N, C)
    dims = head.mask_head[-1].make_dims(B, N, H, W)
    packs = [h.stage_pack(dev) for h in head.mask_head]
    for p in packs:
        p.ensure_prepared(dims)
    fp = torch.zeros(1, N, C, device=dev)
    torch.cuda.synchronize()
    for K in (1, 3, 4, 6, 1, 3):
        sts = [torch.cuda.Stream(device=dev) for _ in range(K)]
        keep = [None] * K
        def run(n):
            for i in range(n):
                with torch.cuda.stream(sts[i % K]):
                    keep[i % K] = vkn.ops.head_forward(dims, packs, x, pfr, mp, None, 4, clip_first_prev=fp)
        run(3 * K + 6)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run(args.steps)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / args.steps
        print(f'B={B} steps in flight on {K} stream(s): {dt * 1e3:7.3f} ms per step  {B / dt:9.1f} frames/s', flush=True)
        del keep, sts
        torch.cuda.empty_cache()


if __name__ == '__main__':
    with torch.no_grad():
        main()

#!/usr/bin/env python3
"""GPU diagnostic (not a test): signed error statistics of gather / decode at BASELINE cfg2 size against fp64 on the GPU.
Looks for accumulation bias (MFMA f16 accumulate truncation) — prints mean(err*sign(ref)), rms, max."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import vkn_import  # noqa: E402

vkn = vkn_import.load()
dev = 'cuda:0'
g = torch.Generator().manual_seed(3)
for (B, N, C, H, W) in [(2, 117, 256, 128, 256), (1, 117, 256, 64, 128), (8, 117, 256, 128, 256)]:
    x = torch.randn(B, C, H, W, generator=g).to(dev)
    for off in (0.0, 1.0):
        xx = x + off                                   # off=1: coherent (non-zero-mean) features like a real network
        m = (torch.randn(B, N, H, W, generator=g) * 4).to(dev)
        k = torch.randn(B, N, C, generator=g).to(dev)
        bits = (m >= vkn.ops.thr_logit(0.5)).double().reshape(B, N, -1)
        for flags in (0, 1):
            xr, cnt = vkn.ops.mask_gather(xx, m, 0.5, flags)
            ref = torch.bmm(bits, xx.double().reshape(B, C, -1).transpose(1, 2))
            e = xr.double() - ref
            print(f'gather  B{B} P{H*W} off{off} flags{flags}: |ref|~{ref.abs().mean():.1f} max|e|={e.abs().max():.3e} '
                  f'rms={e.pow(2).mean().sqrt():.3e} bias(e*sign)={float((e*ref.sign()).mean()):.3e} '
                  f'rel_shrink={float((e*ref).sum()/(ref*ref).sum()):.3e} cnt_ok={bool((cnt.double()==bits.sum(-1)).all())}')
            d = vkn.ops.mask_decode(xx, k, None, flags)
            refd = torch.bmm(k.double(), xx.double().reshape(B, C, -1)).reshape(B, N, H, W)
            e = d.double() - refd
            print(f'decode  B{B} P{H*W} off{off} flags{flags}: |ref|~{refd.abs().mean():.1f} max|e|={e.abs().max():.3e} '
                  f'rms={e.pow(2).mean().sqrt():.3e} bias(e*sign)={float((e*refd.sign()).mean()):.3e} '
                  f'rel_shrink={float((e*refd).sum()/(refd*refd).sum()):.3e}')
        # torch's own fp32 result for comparison (what "a different fp32 summation order" costs)
        t32 = torch.bmm(bits.float(), xx.reshape(B, C, -1).transpose(1, 2))
        e = t32.double() - ref
        print(f'torch32 gather B{B} P{H*W} off{off}: max|e|={e.abs().max():.3e} rms={e.pow(2).mean().sqrt():.3e}')

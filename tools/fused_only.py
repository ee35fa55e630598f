#!/usr/bin/env python3
"""Run ONLY the fused decode->gather pass a few times (for rocprofv3 --pmc runs): python tools/fused_only.py [B] [variant]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import vkn_import  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
if len(sys.argv) > 2:
    os.environ['VKN_FUSED'] = sys.argv[2]
vkn = vkn_import.load()
vkn._lib.use_debug()
dev = torch.device('cuda', 0)
N, C, H, W = 117, 256, 128, 256
g = torch.Generator().manual_seed(0)
x = torch.randn(B, C, H, W, generator=g).to(dev)
kern = (torch.randn(B, N, C, generator=g) * 0.25).to(dev)
kb = torch.randn(B, N, generator=g).to(dev)
hi, lo = vkn.ops.split_planes(kern)
for _ in range(6):
    r = vkn.ops.decode_gather(x, hi, lo, N, kb)
torch.cuda.synchronize()
print('ok', float(r[1].sum()))

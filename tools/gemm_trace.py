#!/usr/bin/env python3
"""Run under rocprofv3 --kernel-trace: library GEMM kernel at several (M, K, N) to read true kernel durations from the trace."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import vkn_import  # noqa: E402

vkn = vkn_import.load()
dev = 'cuda:0'
shapes = [(32, 32, 256), (936, 32, 256), (32, 256, 256), (936, 256, 256), (936, 64, 256), (936, 128, 256), (936, 256, 32)]
for (M, K, N) in shapes:
    A = torch.randn(M, K, device=dev)
    W = torch.randn(N, K, device=dev) / K ** 0.5
    b = torch.randn(N, device=dev)
    ws = vkn.ops.split_weight(W)
    for _ in range(6):
        vkn.ops.linear(A, W, b, ws, 0, 1)
    torch.cuda.synchronize()
    x = torch.zeros(64, device=dev)
    x.add_(float(M * 1000000 + K * 1000 + N))      # marker kernel between shapes
    torch.cuda.synchronize()

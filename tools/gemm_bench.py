#!/usr/bin/env python3
"""GPU diagnostic: time/verify the library GEMM kernel (fp32-MFMA vs bf16x3-split) over K and Nout at M = 936."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import vkn_import  # noqa: E402

vkn = vkn_import.load()
dev = 'cuda:0'


def timeit(fn, reps=50):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for M in [int(a) for a in (sys.argv[1:] or ['936'])]:
  for (K, N, ks) in [(32, 256, 1), (64, 256, 1), (128, 256, 1), (256, 256, 1), (256, 512, 1), (256, 768, 1), (256, 2048, 1),
                     (2048, 256, 8), (256, 32, 1), (256, 19, 1)]:
      A = torch.randn(M, K, device=dev)
      W = torch.randn(N, K, device=dev) / K ** 0.5
      b = torch.randn(N, device=dev)
      ws = vkn.ops.split_weight(W)
      ref = (A.double() @ W.double().t() + b.double())
      o32 = vkn.ops.linear(A, W, b, None, 0, ks)
      os3 = vkn.ops.linear(A, W, b, ws, 0, ks)
      t32 = timeit(lambda: vkn.ops.linear(A, W, b, None, 0, ks))
      ts3 = timeit(lambda: vkn.ops.linear(A, W, b, ws, 0, ks))
      tt = timeit(lambda: torch.addmm(b, A, W.t()))
      print(f'M={M} K={K:5d} N={N:5d} ks={ks}: fp32-mfma {t32:7.1f} us (err {float((o32.double()-ref).abs().max()):.1e})  '
            f'bf16x3 {ts3:7.1f} us (err {float((os3.double()-ref).abs().max()):.1e})  torch.addmm {tt:7.1f} us')
# empty-kernel launch floor through the same path
x = torch.zeros(1024, device=dev)
print(f'torch tiny elementwise launch: {timeit(lambda: x.add_(1.0)):.1f} us/launch (eager, back-to-back)')

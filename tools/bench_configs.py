#!/usr/bin/env python3
"""GPU: frames/s of the fused head at the shapes of BASELINE.json's configs (B = 8 frames per call, fp32, random-init weights,
synthetic inputs), next to bench.py's headline (cfg2).  Informational: parity for these shape classes is covered by tests/."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import vkn_import  # noqa: E402
from test_host_logic import _cfg  # noqa: E402

vkn = vkn_import.load()
dev = torch.device('cuda', 0)
CFGS = [
    # name, video, H, W, N (kernels), nprop, ncls, n_thing, up
    ('cfg1 knet_s3_r50 512x1024 (64x128 feats)', False, 64, 128, 117, 100, 19, 2, 2),
    ('cfg2 video_knet_s3_r50 1024x2048 (128x256)', True, 128, 256, 117, 100, 19, 2, 4),
    ('cfg4 YT-VIS 360x640 pad 384x640 (48x80)', False, 48, 80, 100, 100, 40, 40, 2),
    ('cfg5 VIP-Seg 720p pad 736x1280 (92x160)', True, 92, 160, 166, 100, 124, 58, 4),
]
for name, video, H, W, N, nprop, ncls, nth, up in CFGS:
  for B in (8, 32):
      torch.manual_seed(0)
      head = vkn.build_head(_cfg(video, C=256, heads=8, ffn=2048, ncls=ncls, n_thing=nth, n_stuff=ncls - nth, S=3, up=up,
                                 nprop=nprop))
      head.init_weights()
      head = head.to(dev).eval()
      x = torch.randn(B, 256, H, W, device=dev)
      pf = torch.randn(B, N, 256, device=dev)
      mp = torch.randn(B, N, H, W, device=dev) * 4
      dims = head.mask_head[-1].make_dims(B, N, H, W)
      packs = [h.stage_pack(dev) for h in head.mask_head]
      for want_scaled in (True, False):
          fn = lambda: vkn.ops.head_forward(dims, packs, x, pf, mp, None, up, want_scaled=want_scaled)  # noqa: E731
          for _ in range(5):
              fn()
          e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
          e0.record()
          for _ in range(30):
              fn()
          e1.record()
          torch.cuda.synchronize()
          ms = e0.elapsed_time(e1) / 30
          print(f'{name:48s} x{up} upsample output={str(want_scaled):5s} {ms:7.3f} ms / {B} frames  {B / ms * 1e3:9.0f} frames/s')
      del head, x, pf, mp, packs
      torch.cuda.empty_cache()

#!/usr/bin/env python3
"""bf16-storage head step: live duration of the last-stage decode (events inside the step) vs pixels per workgroup / paired loads
(debug library: VKN_DECODE_PXWG, VKN_DECODE_XPAIR).  One MI355X, cfg2, 32 frames."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tools'))
import vkn_import
vkn = vkn_import.load(); vkn._lib.build_debug(); vkn._lib.use_debug()
import bench
dev = torch.device('cuda', 0)
N, C, H, W, B = 117, 256, 128, 256, 32
head = bench.build_head(vkn, dev)
x, pf, mp = bench.synth_inputs(B, dev, 0)
x = x.to(torch.bfloat16)
pfr = pf.reshape(B, N, C)
dims = head.mask_head[-1].make_dims(B, N, H, W)
packs = [h.stage_pack(dev) for h in head.mask_head]
fp = torch.zeros(1, N, C, device=dev)
alg = B * H * W * (C * 2 + N * 4)
with torch.no_grad():
    for ppw, pair in ((0, 0), (1024, 0), (512, 0), (2048, 0), (0, 1), (1024, 1), (0, 0), (1024, 0)):
        if ppw:
            os.environ['VKN_DECODE_PXWG'] = str(ppw)
        else:
            os.environ.pop('VKN_DECODE_PXWG', None)
        os.environ['VKN_DECODE_XPAIR'] = str(pair)
        for _ in range(8):
            o = vkn.ops.head_forward(dims, packs, x, pfr, mp, None, 4, clip_first_prev=fp)
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(40)]
        for a, b in ev:
            a.record(); b.record()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for a, b in ev:
            o = vkn.ops.head_forward(dims, packs, x, pfr, mp, None, 4, clip_first_prev=fp, decode_events=(a, b))
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / len(ev)
        d = sorted(a.elapsed_time(b) for a, b in ev)
        dm = sum(d) / len(d)
        print(f'px/wg={ppw or "policy":>6} paired={pair}: step {dt * 1e3:6.3f} ms ({B / dt:7.1f} frames/s)  decode live {dm * 1e3:6.1f} us  frac {alg / (dm * 1e-3) / 8e12:.3f}', flush=True)

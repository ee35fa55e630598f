#!/usr/bin/env python3
"""Where to fork the tracking link onto the side stream (debug library, VKN_LINK_FORK_LATE) vs the serial link: step time and the
live duration of the last-stage decode kernel (one MI355X, cfg2, 32 frames)."""
import os, sys, torch, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tools'))
import vkn_import
vkn = vkn_import.load(); vkn._lib.build_debug(); vkn._lib.use_debug()
import bench
dev = torch.device('cuda', 0)
N, C, H, W, B = 117, 256, 128, 256, 32
head = bench.build_head(vkn, dev)
x, pf, mp = bench.synth_inputs(B, dev, 0)
pfr = pf.reshape(B, N, C)
dims = head.mask_head[-1].make_dims(B, N, H, W)
packs = [h.stage_pack(dev) for h in head.mask_head]
fp = torch.zeros(1, N, C, device=dev)
alg = B * H * W * (C + N) * 4
with torch.no_grad():
    for name, env, fl in (('fork at obj_out', '0', 0), ('fork behind the decode', '1', 0), ('serial link', '1', 32), ('fork at obj_out', '0', 0), ('fork behind the decode', '1', 0), ('serial link', '1', 32)):
        os.environ['VKN_LINK_FORK_LATE'] = env
        for _ in range(10):
            o = vkn.ops.head_forward(dims, packs, x, pfr, mp, None, 4, clip_first_prev=fp, flags=fl)
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(60)]
        for a, b in ev:
            a.record(); b.record()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for a, b in ev:
            o = vkn.ops.head_forward(dims, packs, x, pfr, mp, None, 4, clip_first_prev=fp, flags=fl, decode_events=(a, b))
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / len(ev)
        d = sorted(a.elapsed_time(b) for a, b in ev)
        dm = sum(d) / len(d)
        print(f'{name:24s}: step {dt * 1e3:6.3f} ms ({B / dt:7.1f} frames/s)  decode live {dm * 1e3:6.1f} us (min {d[0] * 1e3:.1f} max {d[-1] * 1e3:.1f})  frac {alg / (dm * 1e-3) / 8e12:.3f}', flush=True)

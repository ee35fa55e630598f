#!/usr/bin/env python3
"""Round-3 kernel A/B on one MI355X (debug library: reads VKN_* knobs).  One line per measurement.

    python tools/perf_r03.py --what fused [--frames 32] [--variants 0,1,2,4,7]

fused: k_fused_dgs variant bits (VKN_FUSED_V, csrc/vkn_fused.hip): time of the pass alone in a back-to-back loop, time inside the
       whole head step, and bit-identity of every variant's (xraw, cnt) with variant 0 and with the unfused decode -> gather pair.
chain: the [N x C] update chain alone (vkn_stage_chain_f32) at B = 1 / 8 / 32.
"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def timeit(fn, reps=30, warm=8):
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
    ap.add_argument('--frames', default='32')
    ap.add_argument('--what', default='fused')
    ap.add_argument('--variants', default='0,1,2,4,7')
    ap.add_argument('--shape', default='117,256,128,256', help='N,C,H,W')
    ap.add_argument('--no-head', action='store_true')
    args = ap.parse_args()
    import vkn_import
    vkn = vkn_import.load()
    vkn._lib.build_debug()
    vkn._lib.use_debug()
    import bench
    dev = torch.device('cuda', 0)
    N, C, H, W = (int(v) for v in args.shape.split(','))
    P = H * W
    what = args.what.split(',')
    for B in [int(v) for v in args.frames.split(',')]:
        g = torch.Generator().manual_seed(0)
        x = torch.randn(B, C, H, W, generator=g).to(dev)
        if 'fused' in what:
            kern = (torch.randn(B, N, C, generator=g) * 0.25).to(dev)
            kb = torch.randn(B, N, generator=g).to(dev)
            hi, lo = vkn.ops.split_planes(kern)
            masks = vkn.ops.mask_decode_planes(x, hi, lo, N, kb)
            ref = vkn.ops.mask_gather(x, masks)
            base = None
            for v in [int(t) for t in args.variants.split(',')]:
                # v < 100: k_fused_dgs variant bits; 100 / 101 = k_fused_pp / its profile build; 200 / 201 = k_fused_pq (shipped) / its profile build
                os.environ['VKN_FUSED'] = {100: '3', 101: '4', 200: '5', 201: '6', 202: '7', 203: '8', 204: '9', 300: '10', 301: '11', 302: '10', 400: '12', 401: '13', 402: '14', 403: '15', 404: '16', 305: '17', 306: '18'}.get(v, '2')
                os.environ['VKN_FUSED_CHUNK_LOOP'] = '1' if v == 302 else '0'   # 302: k_fused_il with one launch per 128-row chunk (N > 128)
                os.environ['VKN_FUSED_V'] = str(v)
                out = vkn.ops.decode_gather(x, hi, lo, N, kb)
                torch.cuda.synchronize()
                same_ref = bool(torch.equal(out[0], ref[0]) and torch.equal(out[1], ref[1]))
                if base is None:
                    base = out
                same0 = bool(torch.equal(out[0], base[0]) and torch.equal(out[1], base[1]))
                t = timeit(lambda: vkn.ops.decode_gather(x, hi, lo, N, kb))
                print(f'fused B={B} N={N} C={C} {H}x{W} V={v}: {t:8.1f} us  x-bytes {B * C * P * 4 / t / 1e6:6.2f} TB/s  '
                      f'bit-identical to V=first: {same0}, to decode->gather: {same_ref}', flush=True)
            for pv in [t for t in args.variants.split(',') if t in ('15', '101', '201', '202', '203', '204', '301', '401', '402', '403', '404')]:
                import ctypes
                L = vkn._lib.lib()
                buf = (ctypes.c_ulonglong * 64)()
                os.environ['VKN_FUSED'] = {'15': '2', '101': '4', '201': '6', '202': '7', '203': '8', '204': '9', '301': '11', '401': '13', '402': '14', '403': '15', '404': '16'}[pv]
                os.environ['VKN_FUSED_V'] = '15'
                vkn.ops.decode_gather(x, hi, lo, N, kb)
                torch.cuda.synchronize()
                L.vkn_dbg_fused_prof.argtypes = [ctypes.c_void_p]
                assert L.vkn_dbg_fused_prof(buf) == 0
                ntile = 2 * (((P >> 6) + (256 // B if B <= 256 else 1) - 1) // max(256 // B, 1))
                if pv in ('401', '402', '403', '404'):
                    if pv != '401':
                        print('ABLATION (wrong results): ' + {'402': 'no loader micro-ops in phase D', '403': 'no ballots / writelanes', '404': 'no B-fragment LDS reads after k-step 1'}[pv])
                    print(f'k_fused_w4: cycles per 64-px super-tile, workgroup (0,0), ~{ntile // 2} of them.  slots: 0 phase D (decode + ballots + loader share), 1 barrier A, 2 phase G (gather + loader share), 3 barrier B')
                    for w in range(4):
                        v = [buf[w * 8 + k] / max(ntile // 2, 1) for k in range(4)]
                        print(f'  wave {w}: ' + ' '.join(f'{t:8.0f}' for t in v) + f'   sum {sum(v):8.0f}')
                    continue
                if pv in ('15', '301'):
                    print(f'k_fused_dgs: phase cycles per tile, workgroup (0,0), ~{ntile} tiles: [role work | wait loads | split+LDS write | issue | barrier]')
                    ns = 5
                else:
                    if pv in ('201', '202', '203', '204'):
                        ntile //= 2
                    if pv in ('202', '203', '204'):
                        print('ABLATION (wrong results): ' + {'202': 'gather waves skip their loader share', '203': 'decode waves skip their loader share', '204': 'decode waves read no x fragments from LDS after k-step 1'}[pv])
                    print(f'k_fused_{"pp" if pv == "101" else "pq"}: cycles per {"tile" if pv == "101" else "64-px pair"}, workgroup (0,0), ~{ntile} of them.  slots: 0 -, 1 wait loads, 2 split+LDS write, 3 issue; '
                          'decode waves: 4 decode+ballots, 5 barrier A, 6 barrier B; gather waves: 4 operand prep, 5 barrier A, 6 MFMA issue, 7 barrier B')
                    ns = 8
                for w in range(8):
                    v = [buf[w * 8 + k] / max(ntile, 1) for k in range(ns)]
                    print(f'  wave {w} ({"decode" if w < 4 else "gather"}): ' + ' '.join(f'{t:8.0f}' for t in v) + f'   sum {sum(v):8.0f}')
            os.environ.pop('VKN_FUSED', None)
            if (N, C, H, W) == (117, 256, 128, 256) and not args.no_head:
                head = bench.build_head(vkn, dev)
                xx, pf, mp = bench.synth_inputs(B, dev, 0)
                for v in [int(t) for t in args.variants.split(',')]:
                    os.environ['VKN_FUSED'] = {100: '3', 101: '4', 200: '5', 201: '6', 202: '7', 203: '8', 204: '9', 300: '10', 301: '11', 400: '12', 401: '13'}.get(v, '2')
                    os.environ['VKN_FUSED_V'] = str(v)
                    with torch.no_grad():
                        t = timeit(lambda: head._head_forward(xx, pf, mp, want_scaled=False), reps=20)
                    print(f'head (no upsample) B={B} V={v}: {t:8.1f} us', flush=True)
        if 'chain' in what:
            head = bench.build_head(vkn, dev)
            last = head.mask_head[0]
            dims = last.make_dims(B, N, H, W)
            pack = last.stage_pack(dev)
            xf = torch.randn(B, N, C, device=dev) * 30
            obj = torch.randn(B, N, C, device=dev)
            t = timeit(lambda: vkn.ops.stage_chain(dims, pack, xf, obj))
            print(f'chain B={B}: {t:8.1f} us', flush=True)


if __name__ == '__main__':
    main()

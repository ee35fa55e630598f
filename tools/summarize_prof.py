#!/usr/bin/env python3
"""Condense rocprofv3 (ROCm 7.2 rocpd sqlite) outputs into a text summary for profiles/:
  * per-kernel totals from the kernel trace,
  * the kernel timeline of one timed step,
  * FETCH_SIZE / WRITE_SIZE per kernel from the separate --pmc passes (KB as reported; the MI355X guide's gfx950 caveat:
    FETCH_SIZE reads 1/2 of the bytes of a wide coalesced stream).
usage: summarize_prof.py <prof_dir>"""
import glob
import os
import sqlite3
import sys
from collections import defaultdict

out = sys.argv[1]


def short(n):
    n = n.replace('void ', '')
    if n.startswith('_Z'):
        if 'k_decode_mfma' in n:  # template <NB, ABL, RING, BITS, OPT, XH>: the bit-packed hand-off variant is a different kernel
            import re
            m = re.search(r'k_decode_mfmaILi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)E', n)
            if m and m.group(1) != '4':  # few-row launches (e.g. conv_seg of the kernel-init pass): not the roofline kernel
                return f'k_decode_mfma<NB={m.group(1)}>'
            if m and m.group(6) != '0':  # x stored as fp16 / bf16 (bench.py's x_storage_variants): its own row
                return 'k_decode_mfma<x16>'
            return 'k_decode_mfma<bits>' if (m and m.group(4) == '1') else 'k_decode_mfma'
        if 'k_fused_dgs' in n:
            import re
            m = re.search(r'k_fused_dgsILi(\d+)ELi(\d+)ELi(\d+)E', n)
            return 'k_fused_dgs<x16>' if (m and m.group(3) != '0') else 'k_fused_dgs'
        if 'k_fused_il' in n:   # round 3: the shipped fused pass
            import re
            m = re.search(r'k_fused_ilILi(\d+)ELi(\d+)ELi(\d+)E', n)
            return 'k_fused_il<x16>' if (m and m.group(3) != '0') else 'k_fused_il'
        for key in ('k_chain_a', 'k_chain_c', 'k_chain_pack', 'k_split_planes', 'k_gemm_s3', 'k_fused_dgs', 'k_fused_il', 'k_ffn_fused', 'k_gather_bits_w', 'k_attn_long', 'k_attn_mfma', 'k_attn', 'k_rowepi', 'k_gather_reduce', 'k_upsample_s', 'k_gather_mfma'):
            if key in n:
                return key
    import re
    m = re.match(r'k_fused_il<(\d+), (\d+), (\d+)', n)      # demangled: <NB, C, XH, PF>
    if m:
        return 'k_fused_il<x16>' if m.group(3) != '0' else 'k_fused_il'
    return n.split('(')[0][:48]


def db(tag):
    r = glob.glob(os.path.join(out, tag, '**', '*_results.db'), recursive=True)
    return sqlite3.connect(r[0]) if r else None


con = db('trace')
if con:
    rows = con.execute('select name, start, end from kernels order by start').fetchall()
    # the x-streaming kernels by frames per launch (grid_y): bench.py's breakdown also launches them at 1 / 8 frames
    print('== x-streaming kernels by frames per launch (grid_y) ==')
    for n, gy, c, avg, mn, mx in con.execute(
            "select name, grid_y, count(*), avg(end - start), min(end - start), max(end - start) from kernels where name like "
            "'%k_decode_mfma%' or name like '%k_fused_dgs%' or name like '%k_fused_il%' or name like '%k_gather_mfma%' or name like '%k_upsample_s%' "
            "group by name, grid_y order by name, grid_y"):
        print(f'{short(n):28s} frames/launch={gy:5d} calls={c:4d} avg_us={avg / 1e3:9.2f} min_us={mn / 1e3:9.2f} max_us={mx / 1e3:9.2f}')
    print()
    # the roofline kernel: launches inside head steps (the previous kernel on the timeline is a GEMM) vs bench.py's back-to-back loop
    full_y = max([0] + [gy for n, gy in con.execute("select name, grid_y from kernels where name like '%k_decode_mfma%'")])
    ordered = con.execute('select name, start, end, grid_y from kernels order by start').fetchall()
    instep, loop = [], []
    for i, (n, s0, e0, gy) in enumerate(ordered):
        if short(n) == 'k_decode_mfma' and gy == full_y and i > 0:
            (instep if 'k_decode_mfma' not in ordered[i - 1][0] else loop).append((e0 - s0) / 1e3)
    if instep:
        print(f'k_decode_mfma at {full_y} frames per launch: inside head steps n={len(instep)} avg_us={sum(instep) / len(instep):.2f} '
              f'min={min(instep):.2f} max={max(instep):.2f}; back-to-back loop n={len(loop)} avg_us={sum(loop) / max(len(loop), 1):.2f}')
        # the roofline line recomputed from THIS trace, and the traced run's own bench line beside it (same process: must agree)
        try:
            import json
            import re
            log = open(os.path.join(out, 'bench_trace.log')).read()
            line = json.loads(re.findall(r'^\{.*\}$', log, flags=re.M)[-1])
            rl = line['roofline']
            alg = rl['algorithmic_bytes_per_launch']
            t_in, t_loop = sum(instep) / len(instep), sum(loop) / max(len(loop), 1)
            # bench.py times the last `steps` head steps: the in-step launches of the warm-up / settle steps are in the trace as well
            timed = instep[-line['steps']:]
            t_timed = sum(timed) / len(timed)
            print(f'roofline recomputed from this trace: algorithmic {alg / 1e6:.1f} MB per launch; in-step (all {len(instep)}) '
                  f'{alg / t_in / 1e3 / 8000:.4f}, in-step (the {len(timed)} timed steps) {alg / t_timed / 1e3 / 8000:.4f}, '
                  f'back-to-back loop {alg / t_loop / 1e3 / 8000 if loop else 0:.4f} of 8 TB/s')
            print(f'bench line of the SAME traced process: avg_launch_ms={rl["avg_launch_ms"]} (HIP events) frac={rl["frac"]} '
                  f'isolated_loop_launch_ms={rl["isolated_loop_launch_ms"]} isolated_loop_frac={rl["isolated_loop_frac"]} '
                  f'-> events / trace = {rl["avg_launch_ms"] * 1e3 / t_timed:.3f} (in-step), '
                  f'{rl["isolated_loop_launch_ms"] * 1e3 / t_loop if loop else 0:.3f} (loop)')
        except Exception as e:  # noqa: BLE001
            print('(no bench line beside the trace:', e, ')')
        try:
            d = json.loads(open(os.path.join(out, 'bench_default.json')).read().strip().splitlines()[-1])
            print(f'un-profiled bench line, same box and session: {d["value"]} frames/s, {d["ms_per_step"]} ms per step, decode '
                  f'avg_launch_ms={d["roofline"]["avg_launch_ms"]} frac={d["roofline"]["frac"]} '
                  f'isolated_loop_frac={d["roofline"]["isolated_loop_frac"]}')
        except Exception as e:  # noqa: BLE001
            print('(no un-profiled bench line:', e, ')')
        print()
    agg = defaultdict(lambda: [0, 0.0])
    for n, s, e in rows:
        a = agg[short(n)]
        a[0] += 1
        a[1] += (e - s)
    tot = sum(v[1] for v in agg.values())
    print('== kernel totals (rocprofv3 --kernel-trace) ==')
    print(f'{"kernel":48s} {"calls":>6s} {"avg_us":>9s} {"total_ms":>9s} {"pct":>6s}')
    for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:22]:
        print(f'{k:48s} {c:6d} {t / c / 1e3:9.2f} {t / 1e6:9.3f} {100 * t / tot:6.1f}')
    G = [i for i, r in enumerate(rows) if 'k_gather_mfma' in r[0]]
    if len(G) >= 5:
        # bench.py: every step = one head call = ONE logits gather (stage 0; stages 1.. gather inside the fused pass);
        # take the 4th head call (a timed one)
        i0, j = G[3], G[4]
        t0 = rows[i0][1]
        print('\n== timeline of one timed step (us): start, duration, gap to previous kernel ==')
        prev = None
        gaps = 0.0
        for n, s, e in rows[i0:j]:
            gap = (s - prev) / 1e3 if prev else 0.0
            gaps += max(gap, 0.0)
            print(f'{short(n):32s} t={(s - t0) / 1e3:9.1f} dur={(e - s) / 1e3:8.1f} gap={gap:6.1f}')
            prev = e
        print(f'step span: {(prev - t0) / 1e3:.1f} us, of which gaps {gaps:.1f} us, kernels {j - i0}')

# MFMA utilisation per kernel: SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE x 32 SIMDs per shader engine) — the
# counters are reported per shader engine (8 CUs = 32 SIMDs each); ROCm 7.2 has no gfx950 derived-metric section
mfma_util = {}
con = db('pmc_mfma')
print('\n== MFMA utilisation per kernel (SQ_VALU_MFMA_BUSY_CYCLES / (32 SIMDs x GRBM_GUI_ACTIVE), per shader engine) ==')
if not con:
    print('not collected')
else:
    vals = defaultdict(dict)
    for n, ctr, avg in con.execute('select name, counter_name, avg(counter_value) from pmc_events group by name, counter_name'):
        vals[short(n)][ctr] = vals[short(n)].get(ctr, 0.0) + avg
    for k, v in vals.items():
        if v.get('GRBM_GUI_ACTIVE'):
            mfma_util[k] = v.get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0) / (32.0 * v['GRBM_GUI_ACTIVE'])
    for k, v in sorted(vals.items(), key=lambda kv: -kv[1].get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0))[:12]:
        if v.get('GRBM_GUI_ACTIVE'):
            print(f'{k:48s} MfmaUtil={v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (32.0 * v["GRBM_GUI_ACTIVE"]):6.3f}  '
                  f'SQ_BUSY/GUI_ACTIVE={v.get("SQ_BUSY_CYCLES", 0.0) / v["GRBM_GUI_ACTIVE"]:5.2f}')

pmc = {}
for tag, ctr in (('pmc_fetch', 'FETCH_SIZE'), ('pmc_write', 'WRITE_SIZE')):
    con = db(tag)
    print(f'\n== {ctr} per kernel launch (KB as reported by rocprofv3 --pmc {ctr}) ==')
    if not con:
        print('not collected')
        continue
    for n, c, avg, mx in con.execute('select name, count(*), avg(counter_value), max(counter_value) from pmc_events '
                                     'where counter_name = ? group by name order by avg(counter_value) desc limit 40', (ctr,)):
        # launches of the full batch only (bench.py also launches the kernels at 1 / 8 frames): values within 20 % of the maximum
        # ... and their MEDIAN, so that a single outlier launch (cold first touch: +14 % FETCH_SIZE seen once) neither shifts the
        # figure nor, by raising the maximum, pushes the regular launches out of the window
        vals_ = sorted(v_[0] for v_ in con.execute('select counter_value from pmc_events where counter_name = ? and name = ? and '
                                                   'counter_value >= ?', (ctr, n, 0.8 * mx)))
        full = (len(vals_), vals_[len(vals_) // 2])
        if short(n) in pmc and (ctr + '_KB') in pmc[short(n)]:
            continue  # a second template instance with the same short name (fp16 / bf16): the first (larger) row stands
        if len([1 for v in pmc.values() if (ctr + '_KB') in v]) < 18:
            print(f'{short(n):48s} n={c:5d} mean_KB={avg:14.1f} max_KB={mx:14.1f}  full-batch launches: n={full[0]} median_KB={full[1]:14.1f}')
        pmc.setdefault(short(n), {})[ctr + '_KB'] = full[1]

# sidecar for bench.py's roofline.traffic (committed under profiles/): HBM bytes per launch of the dominant kernels.
# Correction per /opt/skills/guides/MI355X_MICROARCH.md (HBM section): on gfx950 FETCH_SIZE reports 1/2 of the bytes of a wide
# coalesced streaming read -> doubled; WRITE_SIZE taken as reported (it equals the output tensor size exactly here).
if len(sys.argv) > 2 and pmc:
    import json
    side = {}
    for k in ('k_decode_mfma', 'k_decode_mfma<x16>', 'k_fused_il', 'k_fused_il<x16>', 'k_fused_dgs', 'k_fused_dgs<x16>', 'k_gather_mfma', 'k_gather_mfma<4>', 'k_upsample_s', 'k_upsample'):
        if k in pmc and 'FETCH_SIZE_KB' in pmc[k] and 'WRITE_SIZE_KB' in pmc[k]:
            side[k] = dict(fetch_size_kb=pmc[k]['FETCH_SIZE_KB'], write_size_kb=pmc[k]['WRITE_SIZE_KB'],
                           hbm_bytes_per_launch=int((2 * pmc[k]['FETCH_SIZE_KB'] + pmc[k]['WRITE_SIZE_KB']) * 1024))
            if k in mfma_util:  # SQ_VALU_MFMA_BUSY_CYCLES / (32 SIMDs x GRBM_GUI_ACTIVE), all launches of the kernel
                side[k]['mfma_util'] = round(mfma_util[k], 4)
    # matrix-pipe utilisation of every kernel of the step ('MFMA-util per kernel in profiles/r03_pmc.json')
    side['_mfma_util_per_kernel'] = {k: round(v, 4) for k, v in sorted(mfma_util.items(), key=lambda kv: -kv[1]) if v > 0.0005}
    try:  # frames per launch of the profiled run, from the bench line rocprofv3 passed through
        import re
        log = open(os.path.join(out, 'bench_trace.log')).read()
        side['_frames_per_launch'] = int(re.search(r'"frames_per_gpu_per_step": (\d+)', log).group(1))
    except Exception:  # noqa: BLE001
        side['_frames_per_launch'] = None
    import datetime, socket
    side['_collected'] = 'gpurun MI355X box ' + socket.gethostname() + ', ' + datetime.datetime.utcnow().strftime('%Y-%m-%d %H:%M UTC')
    side['_note'] = ('rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes over `bench.py --steps 5 --warmup 2`, '
                     'mean per launch (B = _frames_per_launch frames); bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024 per the gfx950 note in '
                     'MI355X_MICROARCH.md')
    json.dump(side, open(sys.argv[2], 'w'), indent=1)


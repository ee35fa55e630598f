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
        for key in ('k_decode_mfma', 'k_split_planes', 'k_gemm_s3'):
            if key in n:
                return key
    return n.split('(')[0][:48]


def db(tag):
    r = glob.glob(os.path.join(out, tag, '**', '*_results.db'), recursive=True)
    return sqlite3.connect(r[0]) if r else None


con = db('trace')
if con:
    rows = con.execute('select name, start, end from kernels order by start').fetchall()
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
    ups = [i for i, r in enumerate(rows) if 'k_upsample' in r[0]]
    if len(ups) >= 3:
        i0, i1 = ups[-2] + 1, ups[-1] + 1
        # a step = everything after the previous upsample's link kernels .. find first gather after ups[-2]
        g = [i for i in range(ups[-2], ups[-1]) if 'k_gather_mfma' in rows[i][0]]
        i0 = g[0] if g else i0
        t0 = rows[i0][1]
        print('\n== timeline of one timed step (us) ==')
        prev = None
        j = i0
        while j < len(rows) and (j <= ups[-1] or 'k_gather_mfma' not in rows[j][0]):
            n, s, e = rows[j]
            gap = (s - prev) / 1e3 if prev else 0.0
            print(f'{short(n):32s} t={(s - t0) / 1e3:9.1f} dur={(e - s) / 1e3:8.1f} gap={gap:6.1f}')
            prev = e
            j += 1
        print(f'step span: {(prev - t0) / 1e3:.1f} us')

for tag, ctr in (('pmc_fetch', 'FETCH_SIZE'), ('pmc_write', 'WRITE_SIZE')):
    con = db(tag)
    print(f'\n== {ctr} per kernel launch (KB as reported by rocprofv3 --pmc {ctr}) ==')
    if not con:
        print('not collected')
        continue
    for n, c, avg, mx in con.execute('select name, count(*), avg(counter_value), max(counter_value) from pmc_events '
                                     'where counter_name = ? group by name order by avg(counter_value) desc limit 10', (ctr,)):
        print(f'{short(n):48s} n={c:5d} mean_KB={avg:14.1f} max_KB={mx:14.1f}')

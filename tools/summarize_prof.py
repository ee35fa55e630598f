#!/usr/bin/env python3
"""Condense rocprofv3 outputs (kernel stats + per-kernel FETCH_SIZE / WRITE_SIZE) into a small text summary for profiles/."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]


def find(pattern):
    r = glob.glob(os.path.join(out, '**', pattern), recursive=True)
    return r[0] if r else None


def short(n):
    n = n.split('(')[0]
    for p in ('void ', ):
        n = n.replace(p, '')
    return n[:60]


st = find('*kernel_stats.csv')
print('== kernel stats (rocprofv3 --kernel-trace --stats) ==')
if st:
    rows = list(csv.DictReader(open(st)))
    tot = sum(float(r['TotalDurationNs']) for r in rows)
    print(f'{"kernel":60s} {"calls":>6s} {"avg_us":>10s} {"total_ms":>10s} {"pct":>6s}')
    for r in sorted(rows, key=lambda r: -float(r['TotalDurationNs']))[:25]:
        print(f'{short(r["Name"]):60s} {r["Calls"]:>6s} {float(r["AverageNs"]) / 1e3:10.2f} '
              f'{float(r["TotalDurationNs"]) / 1e6:10.3f} {100 * float(r["TotalDurationNs"]) / tot:6.1f}')
else:
    print('no kernel_stats.csv under', out)

for tag, ctr in (('pmc_fetch', 'FETCH_SIZE'), ('pmc_write', 'WRITE_SIZE')):
    f = glob.glob(os.path.join(out, tag, '**', '*counter_collection.csv'), recursive=True)
    print(f'== {ctr} per kernel (KB units as reported; mean over dispatches) ==')
    if not f:
        print('no counter csv')
        continue
    acc = defaultdict(list)
    for r in csv.DictReader(open(f[0])):
        if r.get('Counter_Name') == ctr:
            acc[short(r['Kernel_Name'])].append(float(r['Counter_Value']))
    for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1]))[:12]:
        print(f'{k:60s} n={len(v):5d} mean={sum(v) / len(v):14.1f} max={max(v):14.1f}')

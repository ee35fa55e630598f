"""Per-kernel totals of the LAST steps of a traced `bench.py --train` run (rocprofv3 rocpd sqlite).   usage: train_trace_summary.py <dir> [steps]"""
import collections
import glob
import os
import sqlite3
import sys

out, steps = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 10
f = glob.glob(os.path.join(out, '**', '*_results.db'), recursive=True)[0]
rows = sqlite3.connect(f).execute('select name, start, end from kernels order by start').fetchall()
# the steady-state window: the last 40 % of the kernels; steps in it = window span / the bench line's ms_per_step
import json
ms = None
for line in open(os.path.join(out, 'bench.log')):
    if line.startswith('{'):
        ms = json.loads(line)['ms_per_step']
tail = rows[int(len(rows) * 0.6):]
span = (tail[-1][2] - tail[0][1]) / 1e3
nst = span / (ms * 1e3)
per = len(tail) / nst
agg = collections.defaultdict(lambda: [0, 0.0])
for n, s, e in tail:
    k = n.replace('void ', '').replace('(anonymous namespace)::', '').split('(')[0][:100]
    agg[k][0] += 1
    agg[k][1] += (e - s) / 1e3
tot = sum(v[1] for v in agg.values())
print(f'{per:.0f} kernels per step; window {nst:.1f} steps, {span / nst:.0f} us per step, GPU busy {tot / nst:.0f} us per step ({100 * tot / span:.0f} %)')
for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:int(sys.argv[3]) if len(sys.argv) > 3 else 40]:
    print(f'{t / nst:9.1f} us/step {c / nst:7.1f} x {t / c:8.1f} us  {k}')

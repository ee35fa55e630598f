"""Kernel sequence of the LAST iteration of tools/chain_train_iter.py from its rocprofv3 db.   usage: chain_train_seq.py <dir> <iters + 5>"""
import collections
import glob
import os
import sqlite3
import sys

f = glob.glob(os.path.join(sys.argv[1], '**', '*_results.db'), recursive=True)[0]
n_it = int(sys.argv[2])
rows = sqlite3.connect(f).execute('select name, start, end, grid_x, grid_y from kernels order by start').fetchall()
per = len(rows) // n_it
last = rows[-per:]
t0 = last[0][1]
tot = 0.0
agg = collections.defaultdict(lambda: [0, 0.0])
for n, s, e, gx, gy in last:
    k = n.replace('void ', '').replace('(anonymous namespace)::', '').replace('at::native::', '').split('(')[0][:70]
    if len(sys.argv) > 3:
        print(f'{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:7.1f} us  {k}  grid=({gx},{gy})')
    tot += (e - s) / 1e3
    agg[k][0] += 1
    agg[k][1] += (e - s) / 1e3
print(f'{per} kernels per iteration, span {(last[-1][2] - t0) / 1e3:.0f} us, busy {tot:.0f} us')
for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f'{t:8.1f} us {c:4d} x {t / c:6.1f}  {k}')

#!/bin/bash
# Run ON THE GPU BOX: kernel trace of the training bench step, per-kernel totals + GPU busy fraction.
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/trace_train
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $OUT/trace -o trace -- python $R/bench.py --train --steps 6 --warmup 3 > $OUT/run.log 2>&1
python - <<PY
import sqlite3, glob
from collections import defaultdict
db = glob.glob('$OUT/trace/**/*_results.db', recursive=True)[0]
con = sqlite3.connect(db)
rows = con.execute('select name, start, end from kernels order by start').fetchall()
# last 40 % of the run = timed steps
t0 = rows[0][1]; t1 = rows[-1][2]
cut = t0 + 0.6 * (t1 - t0)
seg = [r for r in rows if r[1] >= cut]
span = seg[-1][2] - seg[0][1]
busy = sum(e - s for _, s, e in seg)
print(f'window {span / 1e6:.1f} ms, kernels {len(seg)}, GPU busy {busy / 1e6:.1f} ms ({100 * busy / span:.0f} %)')
d = defaultdict(lambda: [0, 0.0])
for n, s, e in seg:
    k = n.split('(')[0].replace('void ', '')[:60]
    d[k][0] += 1; d[k][1] += (e - s) / 1e3
for k, v in sorted(d.items(), key=lambda kv: -kv[1][1])[:25]:
    print(f'   {k:62s} {v[0]:5d} {v[1]:9.1f} us')
PY
tail -c 300 $OUT/run.log

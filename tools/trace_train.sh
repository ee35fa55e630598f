#!/bin/bash
# Run ON THE GPU BOX: kernel trace of the training bench step, per-kernel totals + GPU busy fraction.
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/trace_train
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/trace; rocprofv3 --kernel-trace -d /tmp/trace_train -o trace -- python $R/bench.py --train --steps 30 --warmup 10 > $OUT/run.log 2>&1
python - <<PY
import sqlite3, glob
from collections import defaultdict
db = glob.glob('/tmp/trace_train/**/*_results.db', recursive=True)[0]   # (tens of MB: not copied back)
con = sqlite3.connect(db)
rows = con.execute('select name, start, end from kernels order by start').fetchall()
# the last 200 ms before the end of the run = ~18 timed steps in steady state (the TunableOp timing runs are long over)
t1 = rows[-1][2]
seg = [r for r in rows if t1 - 210e6 <= r[1] <= t1 - 10e6]
span = seg[-1][2] - seg[0][1]
busy = sum(e - s for _, s, e in seg)
union, cur_s, cur_e = 0, None, None                      # kernels of a captured graph's parallel branches overlap: union of intervals
for _, s, e in seg:
    if cur_e is None or s > cur_e:
        union += (cur_e - cur_s) if cur_e is not None else 0
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
union += (cur_e - cur_s)
print(f'window {span / 1e6:.1f} ms, kernels {len(seg)}, sum of kernel durations {busy / 1e6:.1f} ms, GPU busy (union) {union / 1e6:.1f} ms ({100 * union / span:.0f} %)')
d = defaultdict(lambda: [0, 0.0])
for n, s, e in seg:
    k = n.split('(')[0].replace('void ', '')[:60]
    d[k][0] += 1; d[k][1] += (e - s) / 1e3
for k, v in sorted(d.items(), key=lambda kv: -kv[1][1])[:25]:
    print(f'   {k:62s} {v[0]:5d} {v[1]:9.1f} us')
PY
tail -c 300 $OUT/run.log

#!/bin/bash
# Run ON THE GPU BOX: kernel trace of the post-head pipeline at cfg2 geometry (tools/panoptic_bench.py + bench.py's panoptic inputs)
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/trace_pan
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/trace_pan
rocprofv3 --kernel-trace --stats -d /tmp/trace_pan -o trace -- python $R/tools/pan_time.py > $OUT/run.log 2>&1
python - <<PY
import sqlite3, glob
from collections import defaultdict
db = glob.glob('/tmp/trace_pan/**/*_results.db', recursive=True)[0]
rows = sqlite3.connect(db).execute('select name, start, end from kernels order by start').fetchall()
d = defaultdict(list)
for n, s, e in rows:
    if 'pan' in n: d[n.split('(')[0]].append((e - s) / 1e3)
for k, v in d.items():
    v = v[len(v) // 2:]
    print(f'{k:40s} n={len(v):4d} avg {sum(v)/len(v):9.1f} us  min {min(v):9.1f}')
PY
tail -3 $OUT/run.log

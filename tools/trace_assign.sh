#!/bin/bash
# Run ON THE GPU BOX: kernel trace of one stage's assignment costs at cfg3 training geometry (tools/assign_lr_time.py)
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/trace_assign
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/trace_assign
rocprofv3 --kernel-trace --stats -d /tmp/trace_assign -o trace -- python $R/tools/assign_lr_time.py "$@" > $OUT/run.log 2>&1
python - <<PY
import sqlite3, glob
from collections import defaultdict
db = glob.glob('/tmp/trace_assign/**/*_results.db', recursive=True)[0]
rows = sqlite3.connect(db).execute('select name, start, end from kernels order by start').fetchall()
d = defaultdict(list)
for n, s, e in rows:
    if 'assign' in n or 'gather' in n: d[n.replace('void ', '').replace('(anonymous namespace)::', '').split('(')[0][:60]].append((e - s) / 1e3)
for k, v in d.items():
    v = v[len(v) // 2:]
    print(f'{k:60s} n={len(v):4d} avg {sum(v)/len(v):9.1f} us  min {min(v):9.1f}')
PY
grep -v amdgpu.ids $OUT/run.log | tail -4

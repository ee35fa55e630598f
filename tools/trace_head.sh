#!/bin/bash
# Run ON THE GPU BOX: kernel trace of the bench step at the given frame counts (tools/perf_r02.py --what head), per-kernel table.
# usage: tools/trace_head.sh <tag> <frames>
TAG=${1:-t}; FR=${2:-1,32}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/trace_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $OUT/trace -o trace -- python $R/tools/perf_r02.py --release --frames $FR --what head > $OUT/run.log 2>&1
python - <<PY
import sqlite3, glob
from collections import defaultdict
db = glob.glob('$OUT/trace/**/*_results.db', recursive=True)[0]
con = sqlite3.connect(db)
rows = con.execute('select name, start, end, grid_y, grid_x from kernels order by start').fetchall()
ups = [i for i, r in enumerate(rows) if 'k_upsample_s' in r[0]]
seen = set()
for a, b in zip(ups, ups[1:]):
    planes = rows[b][3]
    if planes in seen or rows[a][3] != planes: continue
    if sum(1 for u in ups if rows[u][3] == planes and u <= a) < 5: continue   # a warmed-up step
    seen.add(planes)
    seg = rows[a + 1:b + 1]
    print(f'== step with {planes} planes: {len(seg)} kernels, span {(seg[-1][2] - seg[0][1]) / 1e3:.1f} us')
    d = defaultdict(lambda: [0, 0.0])
    for n, s, e, gy, gx in seg:
        k = n.split('(')[0][-44:]
        d[k][0] += 1; d[k][1] += (e - s) / 1e3
    for k, v in sorted(d.items(), key=lambda kv: -kv[1][1]):
        print(f'   {k:46s} {v[0]:3d} {v[1]:9.1f} us  avg {v[1] / v[0]:7.1f}')
    print('   timeline:', ' '.join(f"{n.split('(')[0].replace('void ','')[:12]}:{(e - s) / 1e3:.0f}" for n, s, e, gy, gx in seg))
    t0 = rows[a][2]   # end of the previous step's upsample
    print('   starts (us after the previous step ended: start+duration name grid):')
    for n, s, e, gy, gx in seg:
        print(f"      {(s - t0) / 1e3:8.1f} +{(e - s) / 1e3:7.1f}  {n.split('(')[0].replace('void ','')[:40]}  grid {gx}x{gy}")
PY

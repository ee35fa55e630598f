#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/gemm_trace
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $OUT/t -o t -- python $R/tools/gemm_trace.py > $OUT/log.txt 2>&1
python - <<PY
import sqlite3, glob
con = sqlite3.connect(glob.glob('$OUT/t/**/*_results.db', recursive=True)[0])
rows = con.execute("select name, start, end, grid_x, grid_y from kernels order by start").fetchall()
cur = []
for n,s,e,gx,gy in rows:
    if 'gemm_s3' in n: cur.append((e-s)/1e3)
    elif 'vectorized_elementwise' in n and cur:
        print('gemm_s3 x%d: %s  (grid %s)' % (len(cur), ' '.join('%.1f' % v for v in cur), ''))
        cur = []
PY

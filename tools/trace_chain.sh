#!/bin/bash
# Run ON THE GPU BOX: kernel trace of the [N x C] chain alone (tools/perf_r05.py --what chain) at the given frame counts: per-launch
# durations of the LAST call of every (form, frames) group, in launch order.   usage: tools/trace_chain.sh <tag> <frames> [extra perf_r05 args]
TAG=${1:-c}; FR=${2:-1,8}; shift; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/trace_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $OUT/trace -o trace -- python $R/tools/perf_r05.py --what chain --frames $FR --reps 1 $* > $OUT/run.log 2>&1
cat $OUT/run.log | tail -12
python - <<PY
import sqlite3, glob
db = glob.glob('$OUT/trace/**/*_results.db', recursive=True)[0]
con = sqlite3.connect(db)
rows = con.execute('select name, start, end, grid_x, grid_y, grid_z from kernels order by start').fetchall()
def short(n): return n.split('(')[0].replace('void ', '')[:44]
# a chain call ends with the GEMM that has 2 problems / planes...: split the stream at gaps > 30 us (the timing loops run back to back, so
# instead group by the repeating pattern: find the period of kernel names)
names = [short(r[0]) + f' {r[3]}x{r[4]}x{r[5]}' for r in rows]
i = len(rows) - 1
seen = {}
# walk backwards, print the last occurrence of each distinct consecutive pattern of length <= 16 that starts with a dyn|inp phase
out, last_key = [], None
blocks = []
cur = []
for k, (nm, r) in enumerate(zip(names, rows)):
    cur.append((nm, (r[2] - r[1]) / 1e3, (r[1] - rows[k - 1][2]) / 1e3 if k else 0.0))
    if len(cur) > 1 and cur[-1][2] > 25.0:      # a host-side gap: new timing loop
        blocks.append(cur[:-1]); cur = [cur[-1]]
blocks.append(cur)
for b in blocks:
    if len(b) < 40: continue
    # the period: first index > 0 where the first name reappears with the same following name
    per = next((p for p in range(2, 40) if all(b[j][0] == b[j + p][0] for j in range(0, p))), None)
    if not per: continue
    tail = b[len(b) - per:]
    tot = sum(t for _, t, _ in tail) + sum(g for _, _, g in tail)
    print(f'== pattern of {per} launches, {tot:.1f} us per call (kernel time {sum(t for _, t, _ in tail):.1f} + gaps {sum(g for _, _, g in tail):.1f})')
    for nm, t, gap in tail:
        print(f'     {t:7.2f} us  (gap before {gap:5.2f})  {nm}')
PY

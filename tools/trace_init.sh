#!/bin/bash
# Run ON THE GPU BOX: kernel trace + HBM counters of the kernel-initialisation pass (tools/init_time.py)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/trace_init /tmp/pmc_f /tmp/pmc_w
rocprofv3 --kernel-trace --stats -d /tmp/trace_init -o trace -- python $R/tools/init_time.py > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE -d /tmp/pmc_f -o f -- python $R/tools/init_time.py > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE -d /tmp/pmc_w -o w -- python $R/tools/init_time.py > /dev/null 2>&1
python - <<PY
import sqlite3, glob
from collections import defaultdict
def load(d, q):
    db = glob.glob(d + '/**/*_results.db', recursive=True)[0]
    return sqlite3.connect(db)
con = load('/tmp/trace_init', 0)
rows = con.execute('select name, start, end from kernels order by start').fetchall()
d = defaultdict(list)
for n, s, e in rows: d[n.split('(')[0].replace('void ', '')[:70]].append((e - s) / 1e3)
print('kernel trace (all launches of the run: 2 compare + 13 one-pass + 13 round-5-form calls)')
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
    print(f'  {k:72s} n={len(v):4d} avg {sum(v)/len(v):9.1f} us')
for tag, d_ in (('FETCH_SIZE', '/tmp/pmc_f'), ('WRITE_SIZE', '/tmp/pmc_w')):
    con = load(d_, 0)
    q = con.execute('select name, counter_value from pmc_events where counter_name = ?', (tag,)).fetchall()
    agg = defaultdict(list)
    for n, v in q: agg[n.split('(')[0].replace('void ', '')[:70]].append(v)
    for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1]))[:8]:
        print(f'  {tag} {k:60s} n={len(v):4d} avg {sum(v)/len(v):12.1f} KB' + (f'  (x 2 on gfx950: {2*sum(v)/len(v)/1e3:.1f} MB)' if tag == 'FETCH_SIZE' else f'  ({sum(v)/len(v)/1e3:.1f} MB)'))
PY

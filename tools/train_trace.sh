#!/bin/bash
# Run ON THE GPU BOX (via gpurun): kernel trace of the training bench, per-kernel totals per step.   usage: tools/train_trace.sh <tag> [bench args]
set -u
TAG=${1:-tr}; shift || true
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/train_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
STEPS=10
rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- python $R/bench.py --train --steps $STEPS --warmup 6 $* > $OUT/bench.log 2>&1
tail -1 $OUT/bench.log | cut -c1-200
python - "$OUT" $STEPS <<'PY'
import csv, glob, sys, collections
out, steps = sys.argv[1], int(sys.argv[2])
f = glob.glob(out + '/trace/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# keep the last `steps` steps' worth: take the last 55 % of the kernels (warm-up 6 + capture precede)
n = len(rows)
tail = rows[int(n * 0.6):]
span = (int(tail[-1]['End_Timestamp']) - int(tail[0]['Start_Timestamp'])) / 1e3
agg = collections.defaultdict(lambda: [0, 0.0])
for r in tail:
    k = r['Kernel_Name'].split('(')[0][:90]
    agg[k][0] += 1
    agg[k][1] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
tot = sum(v[1] for v in agg.values())
print(f'kernels in window {len(tail)}, window {span:.0f} us, busy {tot:.0f} us ({100 * tot / span:.0f} %)')
for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:45]:
    print(f'{t:10.0f} us {c:6d} x {t / c:8.1f} us  {k}')
PY

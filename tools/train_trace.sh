#!/bin/bash
# Run ON THE GPU BOX (via gpurun): kernel trace of the training bench, per-kernel totals per step (tools/train_trace_summary.py).
#   usage: tools/train_trace.sh <tag> [bench args]      -> gpurun_out/train_<tag>/{bench.log,summary.txt}; the trace database is deleted
set -u
TAG=${1:-tr}; shift || true
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/train_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- python $R/bench.py --train --steps 10 --warmup 6 "$@" > $OUT/bench.log 2>&1
grep '^{' $OUT/bench.log | tail -1 | cut -c1-200
python $R/tools/train_trace_summary.py $OUT 10 70 | tee $OUT/summary.txt
rm -rf $OUT/trace

#!/bin/bash
# Run ON THE GPU BOX (via gpurun): kernel trace + HBM counters of bench.py; outputs under gpurun_out/.
# usage: tools/gpu_profile.sh <tag> [bench args...]
set -u
TAG=${1:-r02}; shift || true
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# (0) the un-profiled bench line of THIS box, same session (profiled passes run at lower clocks: never compare across the two)
python $R/bench.py --no-cpu-baseline $* > $OUT/bench_default.json 2> $OUT/bench_default.err
ARGS="--steps 20 --warmup 4 --settle 12 --no-cpu-baseline --no-extras $*"
rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- python $R/bench.py $ARGS > $OUT/bench_trace.log 2>&1
# counters in their own passes (FETCH_SIZE and WRITE_SIZE do not fit one pass; never combined with sys-trace)
rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc_fetch -o fetch -- python $R/bench.py $ARGS > $OUT/bench_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc_write -o write -- python $R/bench.py $ARGS > $OUT/bench_write.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $OUT/pmc_mfma -o mfma -- python $R/bench.py $ARGS > $OUT/bench_mfma.log 2>&1
python $R/tools/summarize_prof.py $OUT $OUT/pmc.json > $OUT/summary.txt 2>&1
cat $OUT/summary.txt | head -60
# the trace / counter databases are tens of MB each and gpurun merges at most 64 MiB back: keep the summaries only
rm -rf $OUT/trace $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_mfma

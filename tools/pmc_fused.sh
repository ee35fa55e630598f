#!/bin/bash
# ON THE GPU BOX: PMC passes for the fused decode->gather kernels alone (counters only; never combined with traces).
# usage: tools/pmc_fused.sh <variant 0|1|2> [B]
V=${1:-2}; B=${2:-8}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/pmc_fused_v$V
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_MFMA" \
           "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE" ; do
  i=$((i+1))
  rocprofv3 --pmc $set -d $OUT/p$i -o p$i -- python $R/tools/fused_only.py $B $V > $OUT/p$i.log 2>&1
done
python - <<PY
import sqlite3, glob
for db in sorted(glob.glob('$OUT/p*/*_results.db')):
    con = sqlite3.connect(db); cur = con.cursor()
    try:
        rows = cur.execute("select name, counter_name, count(*), avg(counter_value) from pmc_events where name like '%k_fused%' group by name, counter_name").fetchall()
    except Exception as e:
        print(db, 'ERR', e); continue
    for r in rows: print(f'{r[0][:18]:18s} {r[1]:28s} n={r[2]:3d} avg={r[3]:16.1f}')
PY

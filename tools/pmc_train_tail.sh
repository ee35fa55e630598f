#!/bin/bash
# ON THE GPU BOX: SQ counters of the three x4-sized kernels of the training tail (k_assign_lr, k_ml_fwd_lr, k_ml_bwd_lr) at cfg3 size —
# counters only (never combined with traces).  usage: tools/pmc_train_tail.sh   -> gpurun_out/pmc_train_tail/summary.txt
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/pmc_train_tail
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_MFMA GRBM_GUI_ACTIVE" ; do
  i=$((i+1))
  for prog in assign_lr_time lr_fwd_time lr_bwd_time; do
    rm -rf /tmp/pmc_tt_$i_$prog
    rocprofv3 --pmc $set -d /tmp/pmc_tt_${i}_$prog -o p -- python $R/tools/$prog.py > $OUT/p${i}_$prog.log 2>&1
  done
done
python - > $OUT/summary.txt <<PY
import sqlite3, glob
from collections import defaultdict
acc = defaultdict(lambda: [0, 0.0])
for db in sorted(glob.glob('/tmp/pmc_tt_*/**/*_results.db', recursive=True)):
    con = sqlite3.connect(db); cur = con.cursor()
    try:
        rows = cur.execute("select name, counter_name, count(*), avg(counter_value) from pmc_events where name like '%k_assign_lr%' or name like '%k_ml_fwd_lr%' or name like '%k_ml_bwd_lr%' group by name, counter_name").fetchall()
    except Exception as e:
        print(db, 'ERR', e); continue
    for n, c, k, v in rows:
        n = n.replace('void ', '').replace('(anonymous namespace)::', '').split('(')[0]
        print(f'{n:22s} {c:28s} n={k:3d} avg={v:18.1f}')
PY
cat $OUT/summary.txt
rm -rf /tmp/pmc_tt_*

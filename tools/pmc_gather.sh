#!/bin/bash
# ON THE GPU BOX: PMC passes for the gather kernels over a short bench run (counters only: never combined with traces).
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/pmc_gather
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_INSTS_MFMA SQ_ACTIVE_INST_MISC SQ_WAIT_INST_VMEM" ; do
  i=$((i+1))
  rocprofv3 --pmc $set -d $OUT/p$i -o p$i -- python $R/bench.py --steps 3 --warmup 1 --settle 2 --no-cpu-baseline > $OUT/p$i.log 2>&1
done
python - <<PY
import sqlite3, glob
for db in sorted(glob.glob('$OUT/p*/*_results.db')):
    con = sqlite3.connect(db); cur = con.cursor()
    try:
        rows = cur.execute("select name, counter_name, count(*), avg(counter_value) from pmc_events where name like '%k_gather_mfma%' or name like '%k_decode_mfma%' group by name, counter_name").fetchall()
    except Exception as e:
        print(db, 'ERR', e); continue
    for r in rows:
        short = 'gather<4,1>' if 'ELi1EEv' in r[0] and 'gather' in r[0] else ('gather<4,0>' if 'gather' in r[0] else ('decode_bits' if r[0].rstrip().endswith('Pjf') and 'ELi1EEv' in r[0] else 'decode'))
        print(f'{short:12s} {r[1]:28s} n={r[2]:3d} avg={r[3]:16.1f}')
PY

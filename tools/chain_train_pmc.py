"""MFMA / busy counters per kernel of tools/chain_train_iter.py from a rocprofv3 --pmc run (rocpd sqlite).   usage: chain_train_pmc.py <dir>"""
import collections
import glob
import os
import sqlite3
import sys

f = glob.glob(os.path.join(sys.argv[1], '**', '*_results.db'), recursive=True)[0]
con = sqlite3.connect(f)
vals = collections.defaultdict(dict)
for n, ctr, avg, cnt in con.execute('select name, counter_name, avg(counter_value), count(*) from pmc_events group by name, counter_name'):
    k = n.replace('void ', '').replace('(anonymous namespace)::', '').split('(')[0][:60]
    vals[k][ctr] = avg
    vals[k]['_n'] = cnt
print('MfmaUtil = SQ_VALU_MFMA_BUSY_CYCLES / (32 SIMDs x GRBM_GUI_ACTIVE) per shader engine; GUI_ACTIVE in cycles per launch')
for k, v in sorted(vals.items(), key=lambda kv: -kv[1].get('GRBM_GUI_ACTIVE', 0.0))[:24]:
    ga = v.get('GRBM_GUI_ACTIVE', 0.0)
    if ga:
        print(f'{k:60s} gui_active={ga:10.0f}  MfmaUtil={v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (32.0 * ga):6.3f}  SQ_BUSY/GUI={v.get("SQ_BUSY_CYCLES", 0.0) / ga:6.2f}'
              f'  waves={v.get("SQ_WAVES", 0.0):8.0f}')

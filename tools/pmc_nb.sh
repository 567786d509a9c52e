#!/bin/bash
export TMPDIR=/tmp
ROOT=$(pwd)
for tiles in "" "REMD_NB_TILES=1"; do      # cluster-pair lists, then the 64-atom tile kernel (the fall-back)
dbs=""
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_WAVES"; do
  i=$((i+1))
  (cd /tmp && rm -rf /tmp/pmc_$i && env REMD_OVERLAP=0 $tiles rocprofv3 --pmc $set -d /tmp/pmc_$i -o p -- python $ROOT/tools/small_r_profile.py 24 > /dev/null 2>&1)
  dbs="$dbs $(find /tmp/pmc_$i -name '*.db' | head -1)"
done
echo "== ${tiles:-cluster-pair lists}"
python - $dbs <<'PY'
import sqlite3, sys
tab = {}
for p in sys.argv[1:]:
    db = sqlite3.connect(p)
    for name, ctr, n, avg in db.execute("select kernel_name, counter_name, count(*), avg(value) from counters_collection group by kernel_name, counter_name"):
        tab.setdefault(name.split('(')[0][:48], {})[ctr] = avg
cols = ['SQ_INSTS_VALU', 'SQ_INSTS_SALU', 'SQ_INSTS_LDS', 'SQ_INSTS_VMEM', 'SQ_INSTS_SMEM', 'SQ_WAVES', 'SQ_ACTIVE_INST_VALU', 'SQ_BUSY_CYCLES', 'SQ_WAVE_CYCLES']
print('| kernel | ' + ' | '.join(c.replace('SQ_', '') for c in cols) + ' |')
for k, v in sorted(tab.items(), key=lambda kv: -kv[1].get('SQ_INSTS_VALU', 0)):
    if 'nonbonded' not in k and 'sci' not in k and 'scatter' not in k: continue
    print('| %s | ' % k + ' | '.join('%.4g' % v.get(c, float('nan')) for c in cols) + ' |')
PY
done

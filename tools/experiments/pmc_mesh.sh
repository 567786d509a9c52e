#!/bin/bash
# usage (GPU box, repo root): tools/pmc_mesh.sh  -> instruction counters per dispatch of the mesh kernels, single-launch vs three launches
export TMPDIR=/tmp
ROOT=$(pwd)
for m in 1 0; do
dbs=""
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_WAVES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INST_CYCLES_VMEM"; do
  i=$((i+1))
  (cd /tmp && rm -rf /tmp/pmcm_$i && env REMD_OVERLAP=0 REMD_PME_MESH1=$m rocprofv3 --pmc $set -d /tmp/pmcm_$i -o p -- python $ROOT/tools/small_r_profile.py 24 > /dev/null 2>&1)
  dbs="$dbs $(find /tmp/pmcm_$i -name '*.db' | head -1)"
done
echo "== REMD_PME_MESH1=$m"
python - $dbs <<'PY'
import sqlite3, sys
tab = {}
for p in sys.argv[1:]:
    db = sqlite3.connect(p)
    for name, ctr, n, avg in db.execute("select kernel_name, counter_name, count(*), avg(value) from counters_collection group by kernel_name, counter_name"):
        tab.setdefault(name.split('(')[0][:40], {})[ctr] = avg
cols = ['SQ_INSTS_VALU', 'SQ_INSTS_SALU', 'SQ_INSTS_LDS', 'SQ_INSTS_VMEM', 'SQ_INSTS_SMEM', 'SQ_WAVES', 'SQ_ACTIVE_INST_VALU', 'SQ_BUSY_CYCLES', 'SQ_WAVE_CYCLES', 'SQ_LDS_BANK_CONFLICT', 'SQ_LDS_IDX_ACTIVE', 'SQ_ACTIVE_INST_LDS', 'SQ_WAIT_INST_ANY', 'SQ_WAIT_ANY', 'SQ_INST_CYCLES_VMEM']
print('| kernel | ' + ' | '.join(c.replace('SQ_', '') for c in cols) + ' |')
print('|---|' + '---|' * len(cols))
for k, v in sorted(tab.items(), key=lambda kv: -kv[1].get('SQ_INSTS_VALU', 0)):
    if not k.startswith(('pme_', 'void pme_')): continue
    print('| %s | ' % k + ' | '.join('%.3g' % v.get(c, float('nan')) for c in cols) + ' |')
PY
done

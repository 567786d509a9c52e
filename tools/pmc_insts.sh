#!/bin/bash
# usage (GPU box, repo root): tools/pmc_insts.sh  -> per-kernel instruction counters per dispatch (stand-alone kernels)
export TMPDIR=/tmp
ROOT=$(pwd)
mkdir -p gpurun_out/pmc
dbs=""
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32" "SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_WAVES"; do
  i=$((i+1))
  (cd /tmp && rm -rf /tmp/pmc_$i && env REMD_OVERLAP=0 rocprofv3 --pmc $set -d /tmp/pmc_$i -o p -- python $ROOT/tools/small_r_profile.py 24 > /dev/null 2>&1)
  dbs="$dbs $(find /tmp/pmc_$i -name '*.db' | head -1)"
done
python - $dbs <<'PY'
import sqlite3, sys
tab = {}
for p in sys.argv[1:]:
    db = sqlite3.connect(p)
    for name, ctr, n, avg in db.execute("select kernel_name, counter_name, count(*), avg(value) from counters_collection group by kernel_name, counter_name"):
        tab.setdefault(name.split('(')[0][:48], {})[ctr] = avg
cols = ['SQ_INSTS_VALU', 'SQ_INSTS_VALU_INT32', 'SQ_INSTS_VALU_FMA_F32', 'SQ_INSTS_VALU_TRANS_F32', 'SQ_INSTS_SALU', 'SQ_INSTS_LDS', 'SQ_INSTS_VMEM', 'SQ_INSTS_SMEM', 'SQ_WAVES', 'SQ_ACTIVE_INST_VALU', 'SQ_BUSY_CYCLES', 'SQ_WAVE_CYCLES']
print('| kernel | ' + ' | '.join(c.replace('SQ_', '') for c in cols) + ' |')
print('|---|' + '---|' * len(cols))
for k, v in sorted(tab.items(), key=lambda kv: -kv[1].get('SQ_INSTS_VALU', 0)):
    if v.get('SQ_INSTS_VALU', 0) < 1e4: continue
    print('| %s | ' % k + ' | '.join('%.3g' % v.get(c, float('nan')) for c in cols) + ' |')
PY

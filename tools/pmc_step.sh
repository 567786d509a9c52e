#!/bin/bash
# usage (GPU box, repo root): tools/pmc_step.sh <out.md>  -> SQ instruction / cycle counters per dispatch of every kernel of an MD step
# (24 x alanine dipeptide, REMD_OVERLAP=0: one stream, so a kernel's counters are its own)
export TMPDIR=/tmp
ROOT=$(pwd)
out=${1:-gpurun_out/pmc_step.md}
dbs=""
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_WAVES" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT"; do
  i=$((i+1))
  (cd /tmp && rm -rf /tmp/pmcs_$i && env REMD_OVERLAP=0 rocprofv3 --pmc $set -d /tmp/pmcs_$i -o p -- python $ROOT/tools/small_r_profile.py 24 > /dev/null 2>&1)
  dbs="$dbs $(find /tmp/pmcs_$i -name '*.db' | head -1)"
done
python - $dbs > $out <<'PY'
import sqlite3, sys
tab = {}
for p in sys.argv[1:]:
    db = sqlite3.connect(p)
    for name, ctr, n, avg in db.execute("select kernel_name, counter_name, count(*), avg(value) from counters_collection group by kernel_name, counter_name"):
        t = tab.setdefault(name.split('(')[0][:52], {}); t[ctr] = avg; t['n'] = n
cols = ['SQ_INSTS_VALU', 'SQ_INSTS_SALU', 'SQ_INSTS_LDS', 'SQ_INSTS_VMEM', 'SQ_INSTS_SMEM', 'SQ_WAVES', 'SQ_ACTIVE_INST_VALU', 'SQ_BUSY_CYCLES', 'SQ_WAVE_CYCLES', 'SQ_WAIT_INST_ANY', 'SQ_WAIT_ANY', 'SQ_LDS_BANK_CONFLICT']
print('| kernel | dispatches | ' + ' | '.join(c.replace('SQ_', '') for c in cols) + ' |')
print('|---|---|' + '---|' * len(cols))
for k, v in sorted(tab.items(), key=lambda kv: -kv[1].get('SQ_INSTS_VALU', 0) * kv[1].get('n', 0)):
    if v.get('n', 0) < 100: continue
    print('| %s | %d | ' % (k, v['n']) + ' | '.join('%.3g' % v.get(c, float('nan')) for c in cols) + ' |')
PY
cat $out | cut -c1-200

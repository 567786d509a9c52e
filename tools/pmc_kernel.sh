#!/bin/bash
# usage (GPU box, repo root): tools/pmc_kernel.sh <kernel-name-substring> <python script and args...>  -> SQ counters per dispatch of one kernel
export TMPDIR=/tmp
ROOT=$(pwd)
pat=$1; shift
dbs=""
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVES" "SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_INT32"; do
  i=$((i+1))
  (cd /tmp && rm -rf /tmp/pk_$i && rocprofv3 --pmc $set -d /tmp/pk_$i -o p -- python $ROOT/"$@" > /dev/null 2>&1)
  dbs="$dbs $(find /tmp/pk_$i -name '*.db' | head -1)"
done
python - "$pat" $dbs <<'PY'
import sqlite3, sys
pat = sys.argv[1]
tab = {}
for p in sys.argv[2:]:
    db = sqlite3.connect(p)
    for name, ctr, n, avg in db.execute("select kernel_name, counter_name, count(*), avg(value) from counters_collection group by kernel_name, counter_name"):
        if pat in name:
            tab.setdefault(name.split('(')[0][:50], {})[ctr] = avg
for k, v in tab.items():
    print(k)
    for c, x in sorted(v.items()):
        print('   %-28s %.4g' % (c, x))
PY

#!/bin/bash
# usage (GPU box, repo root): tools/pmc_icache.sh <kernel-name-substring> <python script and args...>  -> instruction-cache counters per dispatch
export TMPDIR=/tmp
ROOT=$(pwd)
pat=$1; shift
dbs=""
i=0
for set in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES" "SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES" "SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_MISC SQ_BUSY_CYCLES"; do
  i=$((i+1))
  (cd /tmp && rm -rf /tmp/pi_$i && rocprofv3 --pmc $set -d /tmp/pi_$i -o p -- python $ROOT/"$@" > /dev/null 2>&1)
  dbs="$dbs $(find /tmp/pi_$i -name '*.db' | head -1)"
done
python - "$pat" $dbs <<'PY'
import sqlite3, sys
pat = sys.argv[1]
tab = {}
for p in sys.argv[2:]:
    db = sqlite3.connect(p)
    for name, ctr, n, avg in db.execute("select kernel_name, counter_name, count(*), avg(value) from counters_collection group by kernel_name, counter_name"):
        if any(q in name for q in pat.split(',')):
            tab.setdefault(name.split('(')[0][:50], {})[ctr] = avg
for k, v in tab.items():
    print(k)
    for c, x in sorted(v.items()):
        print('   %-28s %.4g' % (c, x))
PY

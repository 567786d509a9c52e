#!/bin/bash
# PMC counters + kernel-trace duration of the stand-alone matrix-core XY pass (tools/dft_mfma_bench.py)
export TMPDIR=/tmp
ROOT=$(pwd)
(cd /tmp && rm -rf /tmp/kt_dft && rocprofv3 --kernel-trace --stats -d /tmp/kt_dft -o kt -- python $ROOT/tools/dft_mfma_bench.py "$@" > /dev/null 2>&1)
python - $(find /tmp/kt_dft -name '*.db' | head -1) <<'PY'
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
for name, n, avg, mn in db.execute("select name, count(*), avg(end-start), min(end-start) from kernels group by name"):
    print('%-50s n=%d avg %.1f us min %.1f us' % (name[:50], n, avg / 1e3, mn / 1e3))
PY
dbs=""; i=0
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_INSTS_MFMA SQ_ACTIVE_INST_ANY SQ_WAVES"; do
  i=$((i+1))
  (cd /tmp && rm -rf /tmp/pmcd_$i && rocprofv3 --pmc $set -d /tmp/pmcd_$i -o p -- python $ROOT/tools/dft_mfma_bench.py "$@" > /dev/null 2>&1)
  dbs="$dbs $(find /tmp/pmcd_$i -name '*.db' | head -1)"
done
python - $dbs <<'PY'
import sqlite3, sys
tab = {}
for p in sys.argv[1:]:
    try:
        db = sqlite3.connect(p)
        for name, ctr, n, avg in db.execute("select kernel_name, counter_name, count(*), avg(value) from counters_collection group by kernel_name, counter_name"):
            tab.setdefault(name.split('(')[0][:40], {})[ctr] = avg
    except Exception as e:
        print('pmc pass failed', p, e)
for k, v in tab.items():
    print(k)
    for c, x in sorted(v.items()):
        print('   %-28s %.4g' % (c, x))
PY

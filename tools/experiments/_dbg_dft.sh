export TMPDIR=/tmp
ROOT=$(pwd)
(cd /tmp && rm -rf /tmp/kt_dft && rocprofv3 --kernel-trace -d /tmp/kt_dft -o kt -- python $ROOT/tools/dft_mfma_bench.py 888 75 3 > /dev/null 2>&1)
python - $(find /tmp/kt_dft -name '*.db' | head -1) <<'PY'
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
for name, n, avg, mn in db.execute("select name, count(*), avg(end-start), min(end-start) from kernels where name like 'pme_xy%' group by name"):
    print('avg %.1f us min %.1f us' % (avg / 1e3, mn / 1e3))
PY

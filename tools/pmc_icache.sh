#!/bin/bash
# usage (GPU box, repo root): tools/pmc_icache.sh <out.md>
# Instruction-cache behaviour of every kernel of the headline step (24 x alanine dipeptide, product split, two streams): the step launches
# ~10 different kernels in a row on the same CUs, so each may find the 64 KB instruction cache (shared by two CUs) holding another kernel's
# code.  One rocprofv3 --pmc pass per counter group (never combined with a trace); unknown groups are skipped.
export TMPDIR=/tmp
ROOT=$(pwd)
out=${1:-gpurun_out/pmc_icache.md}
dbs=""
i=0
for set in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_IFETCH SQ_WAIT_INST_ANY" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_ANY" "SQ_IFETCH_LEVEL SQ_ACTIVE_INST_ANY"; do
  i=$((i+1))
  (cd /tmp && rm -rf /tmp/pic_$i && env REMD_TOOLS_EWALD_SPLIT=auto rocprofv3 --pmc $set -d /tmp/pic_$i -o p -- python $ROOT/tools/small_r_profile.py 24 > /tmp/pic_$i.log 2>&1)
  db=$(find /tmp/pic_$i -name '*.db' 2>/dev/null | head -1)
  if [ -n "$db" ]; then dbs="$dbs $db"; else echo "(counter group skipped: $set)"; tail -3 /tmp/pic_$i.log; fi
done
python - $dbs > $out <<'PY'
import sqlite3, sys, re
tab = {}
for p in sys.argv[1:]:
    db = sqlite3.connect(p)
    for name, ctr, n, avg in db.execute("select kernel_name, counter_name, count(*), avg(value) from counters_collection group by kernel_name, counter_name"):
        short = re.sub(r'\(.*', '', name)[:44]
        tab.setdefault(short, {})[ctr] = avg; tab[short]['n'] = n
ctrs = sorted({c for v in tab.values() for c in v if c != 'n'})
print('per-dispatch averages, headline step kernels (24 x alanine dipeptide)')
print('%-44s %6s ' % ('kernel', 'n') + ' '.join('%14s' % c.replace('SQC_ICACHE_', 'IC_').replace('SQ_', '')[:14] for c in ctrs))
for k in sorted(tab, key=lambda k: -tab[k].get('SQ_BUSY_CYCLES', 0)):
    if tab[k]['n'] < 20: continue
    print('%-44s %6d ' % (k, tab[k]['n']) + ' '.join('%14.4g' % tab[k].get(c, float('nan')) for c in ctrs))
PY
cat $out

#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r04_c
mkdir -p $O
timeout 900 python -m pytest tests/test_mix_parity.py tests/test_reference_golden.py -m gpu -x -q > $O/pytest_mix.log 2>&1; tail -3 $O/pytest_mix.log
for pre in 1 0; do
  echo "== REMD_MIX_PRE=$pre" >> $O/mix.txt
  REMD_MIX_PRE=$pre REMD_MIX_FLOW=0 REMD_MIX_DEBUG=1 timeout 300 python tools/mix_microbench.py 24 64 128 192 >> $O/mix.txt 2>&1
done
grep -v amdgpu $O/mix.txt
# DHFR with the 128^2 plane fused
for sp in reference auto; do timeout 300 python tools/split_sweep.py $sp 16 dhfr >> $O/dhfr.txt 2>&1; done
timeout 300 python tools/split_sweep.py auto 16 dhfr standalone >> $O/dhfr.txt 2>&1
timeout 300 python tools/split_sweep.py reference 16 dhfr standalone >> $O/dhfr.txt 2>&1
grep -v amdgpu $O/dhfr.txt
# LDS counters of the mesh kernels and the pair kernel at the auto split
ROOT=$(pwd)
dbs=""
i=0
for set in "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM" "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT" "SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_INST_LEVEL_LDS"; do
  i=$((i+1))
  (cd /tmp && rm -rf /tmp/pl_$i && env REMD_OVERLAP=0 REMD_TOOLS_EWALD_SPLIT=auto rocprofv3 --pmc $set -d /tmp/pl_$i -o p -- python $ROOT/tools/small_r_profile.py 24 > /dev/null 2>&1)
  dbs="$dbs $(find /tmp/pl_$i -name '*.db' | head -1)"
done
python - $dbs > $O/pmc_lds.md <<'PY'
import sqlite3, sys
tab = {}
cols = []
for p in sys.argv[1:]:
    db = sqlite3.connect(p)
    for name, ctr, n, avg in db.execute("select kernel_name, counter_name, count(*), avg(value) from counters_collection group by kernel_name, counter_name"):
        t = tab.setdefault(name.split('(')[0][:44], {}); t[ctr] = avg; t['n'] = n
        if ctr not in cols: cols.append(ctr)
print('| kernel | n | ' + ' | '.join(c.replace('SQ_', '') for c in cols) + ' |')
for k, v in sorted(tab.items(), key=lambda kv: -kv[1].get('SQ_WAVE_CYCLES', 0) * kv[1].get('n', 0)):
    if v.get('n', 0) < 100: continue
    print('| %s | %d | ' % (k, v['n']) + ' | '.join('%.3g' % v.get(c, float('nan')) for c in cols) + ' |')
PY
cat $O/pmc_lds.md | cut -c1-250

#!/bin/bash
# usage (GPU box, repo root): tools/pmc_pair_attrib.sh <out.md>
# Where the pair kernel's wavefront time goes: SQ cycle / instruction counters per dispatch of nonbonded_sci2_kernel,
# stand-alone (REMD_OVERLAP=0: one stream), 24 x alanine dipeptide at the product's Ewald split.  One small rocprofv3 --pmc pass per
# counter group (never combined with a trace); a group whose counter names this rocprofv3 does not know is skipped.
export TMPDIR=/tmp
ROOT=$(pwd)
out=${1:-gpurun_out/pmc_pair_attrib.md}
dbs=""
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM" "SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_WAVES SQ_INST_CYCLES_SALU" \
           "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_IFETCH" \
           "SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_ADD_F32" "SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_CVT SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM"; do
  i=$((i+1))
  (cd /tmp && rm -rf /tmp/ppa_$i && env REMD_OVERLAP=0 REMD_TOOLS_EWALD_SPLIT=auto rocprofv3 --pmc $set -d /tmp/ppa_$i -o p -- python $ROOT/tools/small_r_profile.py 24 > /tmp/ppa_$i.log 2>&1)
  db=$(find /tmp/ppa_$i -name '*.db' 2>/dev/null | head -1)
  if [ -n "$db" ]; then dbs="$dbs $db"; else echo "(counter group skipped: $set)"; fi
done
python - $dbs > $out <<'PY'
import sqlite3, sys
tab = {}
for p in sys.argv[1:]:
    db = sqlite3.connect(p)
    for name, ctr, n, avg in db.execute("select kernel_name, counter_name, count(*), avg(value) from counters_collection group by kernel_name, counter_name"):
        if 'nonbonded_sci2_kernel' in name and 'false, false' in name:
            tab[ctr] = avg; tab['n'] = n
print('nonbonded_sci2_kernel (force-only, Coulomb + LJ sub-system), stand-alone, %d dispatches; per-dispatch averages' % tab.get('n', 0))
for c in sorted(k for k in tab if k != 'n'):
    print('  %-28s %.4g' % (c, tab[c]))
wc = tab.get('SQ_WAVE_CYCLES')
if wc:
    print('shares of SQ_WAVE_CYCLES (wavefront residency, quad-cycles):')
    for c in ('SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY', 'SQ_ACTIVE_INST_ANY', 'SQ_ACTIVE_INST_VALU', 'SQ_ACTIVE_INST_SCA', 'SQ_ACTIVE_INST_LDS', 'SQ_ACTIVE_INST_VMEM', 'SQ_ACTIVE_INST_MISC', 'SQ_WAIT_INST_LDS'):
        if c in tab: print('  %-28s %5.1f %%' % (c, 100.0 * tab[c] / wc))
PY
cat $out

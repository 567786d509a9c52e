#!/bin/bash
# round 5, call 5: pair kernel held back until the plane pass has started (REMD_PAIR_AFTER_XY) on the three PME systems; the pair step's
# issue floor; alchemy-expression diagnostics
export TMPDIR=/tmp
O=gpurun_out/r05_5; mkdir -p $O
python tools/pair_step_floor.py 2>&1 | grep -v amdgpu.ids | tee $O/pair_step_floor.txt
for sysR in "alanine 24" "alanine 8" "hostguest 8" "dhfr 16"; do set -- $sysR
  for v in 0 1 0 1; do REMD_PAIR_AFTER_XY=$v python tools/split_sweep.py auto $2 $1 2>&1 | tail -1 | cut -c60-220; done
done 2>&1 | tee $O/pair_after_xy.txt
python tools/r05/alch_diag.py 2>&1 | grep -v amdgpu.ids > $O/alch_diag.txt; head -c 6000 $O/alch_diag.txt

#!/bin/bash
# GPU run 1 of round 4: parity of the Ewald split + Coulomb table, then the A/B sweeps (split x table x CU masks)
export TMPDIR=/tmp
mkdir -p gpurun_out/r04_a
O=gpurun_out/r04_a
timeout 600 python -m pytest tests/test_openmm_fixture.py tests/test_forcefield_parity.py tests/test_coulomb_table.py -m gpu -x -q > $O/pytest_split.log 2>&1
tail -5 $O/pytest_split.log
S=$O/sweep.txt
: > $S
run() { env "$@" >> $S 2>&1; }
for sp in reference auto 1.21; do
  for tb in 1 0; do
    run REMD_NB_TABLE=$tb timeout 120 python tools/split_sweep.py $sp 24
  done
done
for sp in reference auto 1.21; do
  run timeout 120 python tools/split_sweep.py $sp 24 alanine standalone
  run REMD_NB_TABLE=0 timeout 120 python tools/split_sweep.py $sp 24 alanine standalone
done
# CU masks (pair stream restricted; mesh stream everywhere or restricted to the complement)
for lay in 0 1; do
  for cp in 64 96 128 160 192; do
    run REMD_CU_LAYOUT=$lay REMD_CU_PAIR=$cp REMD_NB_PERSIST_GRID=0 timeout 120 python tools/split_sweep.py auto 24
  done
  for cp in 96 128; do
    run REMD_CU_LAYOUT=$lay REMD_CU_PAIR=$cp REMD_CU_MESH=$((256-cp)) REMD_NB_PERSIST_GRID=0 timeout 120 python tools/split_sweep.py auto 24
  done
done
run REMD_NB_PERSIST_GRID=0 timeout 120 python tools/split_sweep.py auto 24
run REMD_NB_TUNE_VERBOSE=1 timeout 120 python tools/split_sweep.py auto 24
# other systems
for sys in hostguest; do
  for sp in reference auto; do run timeout 200 python tools/split_sweep.py $sp 8 $sys; done
done
for sp in reference auto; do run timeout 300 python tools/split_sweep.py $sp 16 dhfr; done
cat $S

#!/bin/bash
# pair kernel: clamp + table address as three instructions (A = tree, B = libremd_hip_base.so with -DSCI_TABIDX=0: med3 + shift/mask/add)
export TMPDIR=/tmp
B=$PWD/openmmtools_amd/libremd_hip_base.so
timeout 600 python -m pytest tests/test_forcefield_parity.py tests/test_openmm_fixture.py tests/test_coulomb_table.py -m gpu -x -q 2>&1 | tail -3
for i in 1 2; do
python tools/split_sweep.py auto 24 alanine standalone 2>&1 | tail -1 | cut -c150-330
AB_LIB=$B python tools/split_sweep.py auto 24 alanine standalone 2>&1 | tail -1 | cut -c150-330
done
for i in 1 2; do
python tools/split_sweep.py auto 24 alanine 2>&1 | tail -1 | cut -c80-200
AB_LIB=$B python tools/split_sweep.py auto 24 alanine 2>&1 | tail -1 | cut -c80-200
done

#!/bin/bash
# pair kernel: i atoms in the low lane bits (A = tree, B = libremd_hip_base.so with -DSCI_LANES_IJ=0)
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

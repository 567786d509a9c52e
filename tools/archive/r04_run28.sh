#!/bin/bash
# pair kernel: packed x/y accumulators + no upper clamp before the table (A = tree, B = libremd_hip_base.so without both)
export TMPDIR=/tmp
O=gpurun_out/r04_z; mkdir -p $O
B=$PWD/openmmtools_amd/libremd_hip_base.so
for i in 1 2; do
python tools/split_sweep.py auto 24 alanine standalone 2>&1 | tail -1
AB_LIB=$B python tools/split_sweep.py auto 24 alanine standalone 2>&1 | tail -1
done
for i in 1 2; do
python tools/split_sweep.py auto 24 alanine 2>&1 | tail -1
AB_LIB=$B python tools/split_sweep.py auto 24 alanine 2>&1 | tail -1
done
timeout 600 python -m pytest tests/test_forcefield_parity.py tests/test_openmm_fixture.py tests/test_coulomb_table.py -m gpu -x -q 2>&1 | tail -2

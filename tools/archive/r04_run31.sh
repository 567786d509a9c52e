#!/bin/bash
# same-box A/B of the lane remap (A = tree, B = -DSCI_LANES_IJ=0), full steps, alanine x24 and host-guest x8
export TMPDIR=/tmp
B=$PWD/openmmtools_amd/libremd_hip_base.so
for i in 1 2 3; do
python tools/split_sweep.py auto 24 alanine 2>&1 | tail -1 | cut -c80-200
AB_LIB=$B python tools/split_sweep.py auto 24 alanine 2>&1 | tail -1 | cut -c80-200
done
for i in 1 2; do
python tools/split_sweep.py auto 8 hostguest 2>&1 | tail -1 | cut -c80-200
AB_LIB=$B python tools/split_sweep.py auto 8 hostguest 2>&1 | tail -1 | cut -c80-200
done
python bench.py --no-cpu-baseline 2>/dev/null | head -c 250; echo

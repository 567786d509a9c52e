#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r04_y
mkdir -p $O
S=$O/sweep.txt; : > $S
run() { env "$@" >> $S 2>&1; }
run timeout 120 python tools/split_sweep.py auto 24
run timeout 120 python tools/split_sweep.py auto 24
run timeout 120 python tools/split_sweep.py auto 24 alanine standalone
run timeout 120 python tools/split_sweep.py auto 8 hostguest
grep -v amdgpu $S
timeout 900 python -m pytest tests/test_forcefield_parity.py tests/test_openmm_fixture.py tests/test_npt_gpu.py -m gpu -x -q > $O/pytest_a.log 2>&1; grep -E "passed|failed|rror" $O/pytest_a.log | tail -3

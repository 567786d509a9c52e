#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r04_d
mkdir -p $O
timeout 900 python -m pytest tests/test_mix_parity.py tests/test_forcefield_parity.py tests/test_openmm_fixture.py -m gpu -x -q > $O/pytest_a.log 2>&1; tail -3 $O/pytest_a.log
S=$O/sweep.txt; : > $S
run() { env "$@" >> $S 2>&1; }
run REMD_PME_RADIX8=0 timeout 120 python tools/split_sweep.py auto 24
run timeout 120 python tools/split_sweep.py auto 24
run REMD_PME_XYT=512 timeout 120 python tools/split_sweep.py auto 24
run REMD_PME_XYT=256 timeout 120 python tools/split_sweep.py auto 24
run REMD_PME_RADIX8=0 timeout 120 python tools/split_sweep.py auto 24 alanine standalone
run timeout 120 python tools/split_sweep.py auto 24 alanine standalone
run REMD_PME_XYT=512 timeout 120 python tools/split_sweep.py auto 24 alanine standalone
run REMD_PME_XYT=256 timeout 120 python tools/split_sweep.py auto 24 alanine standalone
run timeout 120 python tools/split_sweep.py 1.21 24
run timeout 120 python tools/split_sweep.py auto 8 hostguest
run REMD_PME_RADIX8=0 timeout 120 python tools/split_sweep.py auto 8 hostguest
run timeout 300 python tools/split_sweep.py auto 16 dhfr
grep -v amdgpu $S

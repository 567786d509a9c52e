#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r04_f
mkdir -p $O
timeout 900 python -m pytest tests/test_forcefield_parity.py tests/test_openmm_fixture.py tests/test_harmonic_parity.py tests/test_mts_parity.py tests/test_work_parity.py -m gpu -x -q > $O/pytest_a.log 2>&1; tail -3 $O/pytest_a.log
S=$O/sweep.txt; : > $S
run() { env "$@" >> $S 2>&1; }
for rep in 1 2; do
run REMD_LISTED_MAIN=0 timeout 120 python tools/split_sweep.py auto 24
run timeout 120 python tools/split_sweep.py auto 24
for v in noearly nopf2 nopack none; do
  run AB_LIB=$(pwd)/openmmtools_amd/libremd_hip_$v.so timeout 120 python tools/split_sweep.py auto 24
done
done
run timeout 120 python tools/split_sweep.py auto 24 alanine standalone
for v in noearly nopf2 nopack none; do
  run AB_LIB=$(pwd)/openmmtools_amd/libremd_hip_$v.so timeout 120 python tools/split_sweep.py auto 24 alanine standalone
done
run timeout 120 python tools/split_sweep.py auto 8 hostguest
run REMD_LISTED_MAIN=0 timeout 120 python tools/split_sweep.py auto 8 hostguest
run timeout 300 python tools/split_sweep.py auto 16 dhfr
run REMD_LISTED_MAIN=0 timeout 300 python tools/split_sweep.py auto 16 dhfr
grep -v amdgpu $S
ROOT=$(pwd)
(cd /tmp && rm -rf /tmp/tl && REMD_TOOLS_EWALD_SPLIT=auto rocprofv3 --kernel-trace -d /tmp/tl -o t -- python $ROOT/tools/small_r_profile.py 24 > /dev/null 2>&1)
python tools/timeline_step.py /tmp/tl 60 2 > $O/timeline.txt; cat $O/timeline.txt

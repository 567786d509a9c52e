#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r04_p
mkdir -p $O
S=$O/sweep.txt; : > $S
run() { env "$@" >> $S 2>&1; }
for rep in 1 2; do
run REMD_NB_FOLD=0 timeout 120 python tools/split_sweep.py auto 24
run timeout 120 python tools/split_sweep.py auto 24
done
run timeout 120 python tools/split_sweep.py auto 8 hostguest
run REMD_NB_FOLD=0 timeout 120 python tools/split_sweep.py auto 8 hostguest
grep -v amdgpu $S
timeout 900 python -m pytest tests/test_forcefield_parity.py tests/test_openmm_fixture.py tests/test_distributed_gpu.py tests/test_harmonic_parity.py -m gpu -x -q > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log
ROOT=$(pwd)
(cd /tmp && rm -rf /tmp/tl && REMD_NB_PRIO=1 REMD_NB_PERSIST_GRID=0 REMD_TOOLS_EWALD_SPLIT=auto rocprofv3 --kernel-trace -d /tmp/tl -o t -- python $ROOT/tools/small_r_profile.py 24 > /dev/null 2>&1)
python tools/timeline_step.py /tmp/tl 60 2 > $O/timeline.txt; cat $O/timeline.txt

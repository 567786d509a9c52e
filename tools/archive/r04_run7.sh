#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r04_g
mkdir -p $O
S=$O/sweep.txt; : > $S
run() { env "$@" >> $S 2>&1; }
for rep in 1 2; do
run REMD_LISTED_MAIN=0 timeout 120 python tools/split_sweep.py auto 24
run timeout 120 python tools/split_sweep.py auto 24
done
run timeout 120 python tools/split_sweep.py auto 8 hostguest
run REMD_LISTED_MAIN=0 timeout 120 python tools/split_sweep.py auto 8 hostguest
run timeout 300 python tools/split_sweep.py auto 16 dhfr
run REMD_LISTED_MAIN=0 timeout 300 python tools/split_sweep.py auto 16 dhfr
grep -v amdgpu $S
ROOT=$(pwd)
(cd /tmp && rm -rf /tmp/tl && REMD_TOOLS_EWALD_SPLIT=auto rocprofv3 --kernel-trace -d /tmp/tl -o t -- python $ROOT/tools/small_r_profile.py 24 > /dev/null 2>&1)
python tools/timeline_step.py /tmp/tl 60 2 > $O/timeline.txt; cat $O/timeline.txt


#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r04_k
mkdir -p $O
ROOT=$(pwd)
(cd /tmp && rm -rf /tmp/tl && REMD_TOOLS_EWALD_SPLIT=auto rocprofv3 --kernel-trace -d /tmp/tl -o t -- python $ROOT/tools/small_r_profile.py 24 > /dev/null 2>&1)
python tools/timeline_step.py /tmp/tl 60 2 > $O/timeline.txt; cat $O/timeline.txt
REMD_PROF_EVERY=64 python bench.py --no-cpu-baseline > $O/bench_prof64.json 2> /dev/null; head -c 250 $O/bench_prof64.json; echo
python bench.py --no-cpu-baseline > $O/bench_prof16.json 2> /dev/null; head -c 250 $O/bench_prof16.json; echo

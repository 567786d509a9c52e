#!/bin/bash
# usage (GPU box, repo root): tools/tl_step.sh <tag> [env assignments...]  -- kernel timeline of two MD steps (24 x alanine dipeptide)
tag=$1; shift
export TMPDIR=/tmp
ROOT=$(pwd)
mkdir -p $ROOT/gpurun_out
(cd /tmp && rm -rf /tmp/tl_$tag && env "$@" rocprofv3 --kernel-trace -d /tmp/tl_$tag -o kt -- python $ROOT/tools/launch_bound_check.py 24 > /dev/null 2>&1)
python tools/timeline2.py /tmp/tl_$tag > gpurun_out/timeline_$tag.txt 2>&1
cat gpurun_out/timeline_$tag.txt

#!/bin/bash
# usage (on the GPU box, repo root): tools/collect_profiles_light.sh <tag>
# the default bench line + the rocprofv3 kernel-trace statistics of the same command (no PMC passes: those take ~6 GPU-minutes)
tag=${1:-r02_x}
export TMPDIR=/tmp
ROOT=$(pwd)
out=$ROOT/gpurun_out/$tag
mkdir -p $out
python bench.py > $out/bench_default.json 2> $out/bench_default.err
(cd /tmp && rm -rf /tmp/prof_kt && rocprofv3 --kernel-trace --stats -d /tmp/prof_kt -o kt -- python $ROOT/bench.py --no-cpu-baseline > $out/bench_under_rocprof.json 2> /dev/null)
python tools/rocpd_stats.py $(find /tmp/prof_kt -name "*.db" | head -1) $out/kernel_stats.md > /dev/null
head -c 400 $out/bench_default.json; echo; head -14 $out/kernel_stats.md | cut -c1-150

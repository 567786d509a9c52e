#!/bin/bash
# usage (on the GPU box, repo root): tools/collect_profiles.sh <tag>
# Produces under gpurun_out/<tag>/: the default bench line, the rocprofv3 kernel-trace stats of the same command, and
# the HBM-side traffic counters (separate --pmc passes, never combined with other traces).
# (round 5: the profiled command is the default one WITHOUT the extra ensemble shapes and the CPU leg -- `--no-shapes --no-cpu-baseline` --
# so that a kernel's average is the headline configuration's; roofline.avg_launch_ms of the bench line is measured inside the timed
# region of the headline configuration whatever runs afterwards)
tag=${1:-r01_x}
export TMPDIR=/tmp
ROOT=$(pwd)
out=$ROOT/gpurun_out/$tag
mkdir -p $out
python bench.py > $out/bench_default.json 2> $out/bench_default.err
(cd /tmp && rm -rf /tmp/prof_kt && rocprofv3 --kernel-trace --stats -d /tmp/prof_kt -o kt -- python $ROOT/bench.py --no-cpu-baseline --no-shapes > $out/bench_under_rocprof.json 2> /dev/null)
python tools/rocpd_stats.py $(find /tmp/prof_kt -name "*.db" | head -1) $out/kernel_stats.md > /dev/null
dbs=""
for c in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && rm -rf /tmp/prof_$c && rocprofv3 --pmc $c -d /tmp/prof_$c -o pmc -- python $ROOT/bench.py --steps 1 --warmup 0 --md-steps 50 --no-cpu-baseline --no-shapes > /dev/null 2>&1)
  dbs="$dbs $(find /tmp/prof_$c -name "*.db" | head -1)"
done
python tools/rocpd_pmc.py $dbs --md $out/pmc_traffic.md --json $out/pmc_traffic.json > /dev/null
# per-kernel roofline table: durations x PMC traffic x algorithmic bytes, against the STREAM triad measured by bench.py
stream=$(python -c "import json,sys; d=json.load(open('$out/bench_default.json')); print(d.get('measured_roofs',{}).get('stream_triad_gb_per_s',0))" 2>/dev/null)
rpl=$(python -c "import json,sys; d=json.load(open('$out/bench_default.json')); print(int(d['roofline'].get('replicas_per_launch', 24)))" 2>/dev/null)
python tools/kernel_roofs.py $out/kernel_stats.md $out/pmc_traffic.json --stream ${stream:-6300} --replicas ${rpl:-24} > $out/kernel_roofs.md
git rev-parse HEAD > /dev/null 2>&1 || true
head -c 600 $out/bench_default.json; echo; head -8 $out/kernel_stats.md | cut -c1-150; head -6 $out/pmc_traffic.md | cut -c1-150

#!/bin/bash
# round 6, call 33: timelines of the final tree (alanine x 24 and DHFR x 16, two phases), kernel statistics of the phased DHFR run
export TMPDIR=/tmp
ROOT=$(pwd); O=$ROOT/gpurun_out/r06_33; mkdir -p $O
tl() { tag=$1; shift
  (cd /tmp && rm -rf /tmp/tl_$tag && env "$@" rocprofv3 --kernel-trace -d /tmp/tl_$tag -o kt -- python $ROOT/tools/phase_probe.py ${ARGS} > $O/run_$tag.txt 2>&1)
  python tools/timeline_window.py /tmp/tl_$tag ${WIN} ${BACK} > $O/timeline_$tag.txt 2>&1; head -2 $O/timeline_$tag.txt
  python tools/rocpd_stats.py $(ls /tmp/tl_$tag/*/*.db /tmp/tl_$tag/*.db 2>/dev/null | head -1) > $O/stats_$tag.txt 2>&1; head -14 $O/stats_$tag.txt | cut -c1-120; }
BACK=150 WIN=400 ARGS="24 1 seq" tl ala_p2 GO_ITERS=2 GO_PHASES=2
BACK=40 WIN=2600 ARGS="16 1 seq dhfr" tl dhfr_p2 GO_STEPS=600 GO_ITERS=1 GO_PHASES=2
BACK=40 WIN=2600 ARGS="16 1 seq dhfr" tl dhfr_p1 GO_STEPS=600 GO_ITERS=1 GO_PHASES=1

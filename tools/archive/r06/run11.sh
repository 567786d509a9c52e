#!/bin/bash
# round 6, call 11: timelines behind the tuner (steps 540+ of 600)
export TMPDIR=/tmp
ROOT=$(pwd); O=$ROOT/gpurun_out/r06_11; mkdir -p $O
tl() { tag=$1; shift
  (cd /tmp && rm -rf /tmp/tl_$tag && env "$@" rocprofv3 --kernel-trace -d /tmp/tl_$tag -o kt -- python $ROOT/tools/phase_probe.py ${ARGS} > $O/run_$tag.txt 2>&1)
  python tools/timeline_window.py /tmp/tl_$tag ${WIN} ${BACK} > $O/timeline_$tag.txt 2>&1; head -2 $O/timeline_$tag.txt; }
BACK=40 WIN=3000 ARGS="16 1 seq dhfr" tl dhfr_p1 GO_STEPS=600 GO_ITERS=1 GO_PHASES=1
BACK=40 WIN=3000 ARGS="16 1 seq dhfr" tl dhfr_p1_pax GO_STEPS=600 GO_ITERS=1 GO_PHASES=1 REMD_PAIR_AFTER_XY=1
BACK=80 WIN=3000 ARGS="16 1 seq dhfr" tl dhfr_p2 GO_STEPS=600 GO_ITERS=1 GO_PHASES=2
BACK=80 WIN=400 ARGS="24 1 seq" tl ala_p2 GO_STEPS=700 GO_ITERS=1 GO_PHASES=2
BACK=40 WIN=400 ARGS="24 1 seq" tl ala_p1 GO_STEPS=700 GO_ITERS=1 GO_PHASES=1

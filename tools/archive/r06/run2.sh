#!/bin/bash
# round 6, call 2: kernel timelines of two 12-replica handles driven from two host threads (flags / events) against one 24-replica handle
export TMPDIR=/tmp
ROOT=$(pwd); O=$ROOT/gpurun_out/r06_2; mkdir -p $O
tl() { tag=$1; shift
  (cd /tmp && rm -rf /tmp/tl_$tag && env GO_STEPS=200 GO_ITERS=2 "$@" rocprofv3 --kernel-trace -d /tmp/tl_$tag -o kt -- python $ROOT/tools/phase_probe.py ${ARGS} > $O/run_$tag.txt 2>&1)
  python tools/timeline_window.py /tmp/tl_$tag 500 150 > $O/timeline_$tag.txt 2>&1; head -3 $O/timeline_$tag.txt; }
ARGS="24 1 seq" tl g1
ARGS="24 2 thr" tl g2_flags
ARGS="24 2 thr" tl g2_events REMD_SYNC_EVENTS=1 REMD_NB_PERSIST_GRID=0
ARGS="12 1 seq" tl r12
ARGS="12 1 seq" tl r12_events REMD_SYNC_EVENTS=1 REMD_NB_PERSIST_GRID=0

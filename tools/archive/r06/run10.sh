#!/bin/bash
# round 6, call 10: DHFR -- the pair kernel held back until the plane pass has ended, with one and two phases; timelines of the phased steps
export TMPDIR=/tmp
ROOT=$(pwd); O=$ROOT/gpurun_out/r06_10; mkdir -p $O
P="python tools/phase_probe.py"
{
env GO_ITERS=8 GO_STEPS=100 GO_PHASES=1 $P 16 1 seq dhfr
env GO_ITERS=8 GO_STEPS=100 GO_PHASES=1 REMD_PAIR_AFTER_XY=1 $P 16 1 seq dhfr
env GO_ITERS=8 GO_STEPS=100 GO_PHASES=2 $P 16 1 seq dhfr
env GO_ITERS=8 GO_STEPS=100 GO_PHASES=2 REMD_PAIR_AFTER_XY=1 $P 16 1 seq dhfr
env GO_ITERS=3 GO_PHASES=1 $P 48 1 seq
env GO_ITERS=3 GO_PHASES=2 $P 48 1 seq
env GO_ITERS=2 GO_STEPS=200 GO_PHASES=1 $P 128 1 seq
env GO_ITERS=2 GO_STEPS=200 GO_PHASES=2 $P 128 1 seq
env GO_ITERS=2 GO_STEPS=200 GO_PHASES=1 $P 64 1 seq
env GO_ITERS=2 GO_STEPS=200 GO_PHASES=2 $P 64 1 seq
} 2>&1 | grep -v "amdgpu.ids\|per-replica" | cut -c1-260 | sed 's/ first .*//' | tee $O/probe.txt
tl() { tag=$1; shift
  (cd /tmp && rm -rf /tmp/tl_$tag && env "$@" rocprofv3 --kernel-trace -d /tmp/tl_$tag -o kt -- python $ROOT/tools/phase_probe.py ${ARGS} > $O/run_$tag.txt 2>&1)
  python tools/timeline_window.py /tmp/tl_$tag ${WIN} 150 > $O/timeline_$tag.txt 2>&1; head -2 $O/timeline_$tag.txt; }
WIN=500 ARGS="24 1 seq" tl ala_p2 GO_STEPS=700 GO_ITERS=1 GO_PHASES=2
WIN=500 ARGS="24 1 seq" tl ala_p1 GO_STEPS=700 GO_ITERS=1 GO_PHASES=1
WIN=3200 ARGS="16 1 seq dhfr" tl dhfr_p1 GO_STEPS=600 GO_ITERS=1 GO_PHASES=1
WIN=3200 ARGS="16 1 seq dhfr" tl dhfr_p1_pax GO_STEPS=600 GO_ITERS=1 GO_PHASES=1 REMD_PAIR_AFTER_XY=1
WIN=3200 ARGS="16 1 seq dhfr" tl dhfr_p2 GO_STEPS=600 GO_ITERS=1 GO_PHASES=2

#!/bin/bash
# round 6, call 22: stand-alone kernel durations (REMD_OVERLAP=0, one block) with the power-of-two mesh passes, and the scheduled ones
export TMPDIR=/tmp
ROOT=$(pwd); O=$ROOT/gpurun_out/r06_22; mkdir -p $O
st() { tag=$1; shift
  (cd /tmp && rm -rf /tmp/st_$tag && env "$@" rocprofv3 --kernel-trace -d /tmp/st_$tag -o kt -- python $ROOT/tools/phase_probe.py ${ARGS} > $O/run_$tag.txt 2>&1)
  python tools/rocpd_stats.py $(ls /tmp/st_$tag/*/*.db /tmp/st_$tag/*.db 2>/dev/null | head -1) > $O/stats_$tag.txt 2>&1; echo "== $tag"; head -14 $O/stats_$tag.txt | cut -c1-150; }
ARGS="24 1 seq" st ala_alone GO_ITERS=1 GO_PHASES=1 REMD_OVERLAP=0
ARGS="24 1 seq" st ala_alone_sched GO_ITERS=1 GO_PHASES=1 REMD_OVERLAP=0 REMD_PME_POW2=0
ARGS="16 1 seq dhfr" st dhfr_alone GO_STEPS=100 GO_ITERS=1 GO_PHASES=1 REMD_OVERLAP=0
ARGS="16 1 seq dhfr" st dhfr_alone_sched GO_STEPS=100 GO_ITERS=1 GO_PHASES=1 REMD_OVERLAP=0 REMD_PME_POW2=0
ARGS="16 1 seq dhfr" st dhfr_p2 GO_STEPS=100 GO_ITERS=2 GO_PHASES=2
ARGS="16 1 seq dhfr" st dhfr_p1 GO_STEPS=100 GO_ITERS=2 GO_PHASES=1

#!/bin/bash
# round 6, call 21: timelines and kernel statistics with the power-of-two mesh passes (alanine x 24 phased / one block, DHFR x 16)
export TMPDIR=/tmp
ROOT=$(pwd); O=$ROOT/gpurun_out/r06_21; mkdir -p $O
tl() { tag=$1; shift
  (cd /tmp && rm -rf /tmp/tl_$tag && env "$@" rocprofv3 --kernel-trace -d /tmp/tl_$tag -o kt -- python $ROOT/tools/phase_probe.py ${ARGS} > $O/run_$tag.txt 2>&1)
  python tools/timeline_window.py /tmp/tl_$tag ${WIN} ${BACK} > $O/timeline_$tag.txt 2>&1; head -2 $O/timeline_$tag.txt
  python tools/rocpd_stats.py /tmp/tl_$tag > $O/stats_$tag.txt 2>&1; head -14 $O/stats_$tag.txt; }
BACK=150 WIN=450 ARGS="24 1 seq" tl ala_p2 GO_ITERS=2 GO_PHASES=2
BACK=150 WIN=450 ARGS="24 1 seq" tl ala_p1 GO_ITERS=2 GO_PHASES=1
BACK=40 WIN=3000 ARGS="16 1 seq dhfr" tl dhfr_p2 GO_STEPS=600 GO_ITERS=1 GO_PHASES=2
REMD_OVERLAP=0 python tools/mesh_standalone.py 24 2>&1 | grep -v amdgpu | tee $O/standalone.txt
REMD_OVERLAP=0 python tools/mesh_standalone.py 16 dhfr 2>&1 | grep -v amdgpu | tee -a $O/standalone.txt

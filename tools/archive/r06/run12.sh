#!/bin/bash
# round 6, call 12: DHFR with the pair kernel behind the plane pass and the listed terms in front of the wait
export TMPDIR=/tmp
ROOT=$(pwd); O=$ROOT/gpurun_out/r06_12; mkdir -p $O
P="python tools/phase_probe.py"
{
env GO_ITERS=8 GO_STEPS=100 GO_PHASES=1 REMD_PAIR_AFTER_XY=0 $P 16 1 seq dhfr
env GO_ITERS=8 GO_STEPS=100 GO_PHASES=1 $P 16 1 seq dhfr
env GO_ITERS=8 GO_STEPS=100 GO_PHASES=2 REMD_PAIR_AFTER_XY=0 $P 16 1 seq dhfr
env GO_ITERS=8 GO_STEPS=100 GO_PHASES=2 $P 16 1 seq dhfr
env GO_ITERS=3 GO_PHASES=1 $P 8 1 seq hostguest
env GO_ITERS=3 GO_PHASES=1 REMD_PAIR_AFTER_XY=1 $P 8 1 seq hostguest
env GO_ITERS=3 GO_PHASES=1 REMD_PAIR_AFTER_XY=1 $P 24 1 seq
} 2>&1 | grep -v "amdgpu.ids\|per-replica" | cut -c1-260 | sed 's/ first .*//' | tee $O/probe.txt
tl() { tag=$1; shift
  (cd /tmp && rm -rf /tmp/tl_$tag && env "$@" rocprofv3 --kernel-trace -d /tmp/tl_$tag -o kt -- python $ROOT/tools/phase_probe.py ${ARGS} > $O/run_$tag.txt 2>&1)
  python tools/timeline_window.py /tmp/tl_$tag ${WIN} ${BACK} > $O/timeline_$tag.txt 2>&1; head -2 $O/timeline_$tag.txt; }
BACK=40 WIN=3000 ARGS="16 1 seq dhfr" tl dhfr_p1_pax GO_STEPS=600 GO_ITERS=1 GO_PHASES=1
timeout 600 python -m pytest tests/test_forcefield_parity.py -m gpu -q -x -k "config5 or dhfr or stream_modes or phases" 2>&1 | tail -4 | tee $O/pytest.txt

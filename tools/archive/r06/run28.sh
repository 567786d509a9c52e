#!/bin/bash
# round 6, call 28: full GPU suite on the tree with the power-of-two mesh passes, listed entries by atom and mesh forces by bin position;
# stand-alone kernel durations of the new tree
export TMPDIR=/tmp
ROOT=$(pwd); O=$ROOT/gpurun_out/r06_28; mkdir -p $O
st() { tag=$1; shift
  (cd /tmp && rm -rf /tmp/st_$tag && env "$@" rocprofv3 --kernel-trace -d /tmp/st_$tag -o kt -- python $ROOT/tools/phase_probe.py ${ARGS} > $O/run_$tag.txt 2>&1)
  python tools/rocpd_stats.py $(ls /tmp/st_$tag/*/*.db /tmp/st_$tag/*.db 2>/dev/null | head -1) > $O/stats_$tag.txt 2>&1; echo "== $tag"; head -15 $O/stats_$tag.txt | cut -c1-110; }
ARGS="24 1 seq" st ala_alone GO_ITERS=1 GO_PHASES=1 REMD_OVERLAP=0
ARGS="16 1 seq dhfr" st dhfr_alone GO_STEPS=100 GO_ITERS=1 GO_PHASES=1 REMD_OVERLAP=0
ARGS="8 1 seq hostguest" st hg_alone GO_STEPS=200 GO_ITERS=1 GO_PHASES=1 REMD_OVERLAP=0
timeout 1700 python -m pytest tests -m gpu -q 2>&1 | tail -6 | tee $O/pytest.txt
python bench.py 2> $O/bench.err | tee $O/bench.json | cut -c1-300

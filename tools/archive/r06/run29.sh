#!/bin/bash
# round 6, call 29: list builder with four wavefronts per tile (same lists, fewer dependent round trips): digests must equal call 27's
# (alanine 04781a1c28, DHFR 2c490e305a / 1852654ed7, host-guest 37a3e779f9)
export TMPDIR=/tmp
ROOT=$(pwd); O=$ROOT/gpurun_out/r06_29; mkdir -p $O
P="python tools/phase_probe.py"
{
env GO_ITERS=4 GO_PHASES=2 $P 24 1 seq
env GO_ITERS=4 GO_PHASES=2 $P 24 1 seq
env GO_ITERS=6 GO_STEPS=100 GO_PHASES=2 $P 16 1 seq dhfr
env GO_ITERS=6 GO_STEPS=100 GO_PHASES=1 $P 16 1 seq dhfr
env GO_ITERS=3 GO_PHASES=1 $P 8 1 seq hostguest
} 2>&1 | grep -v "amdgpu.ids\|per-replica\|host enqueue" | cut -c1-260 | sed 's/ first .*//' | tee $O/probe.txt
st() { tag=$1; shift
  (cd /tmp && rm -rf /tmp/st_$tag && env "$@" rocprofv3 --kernel-trace -d /tmp/st_$tag -o kt -- python $ROOT/tools/phase_probe.py ${ARGS} > $O/run_$tag.txt 2>&1)
  python tools/rocpd_stats.py $(ls /tmp/st_$tag/*/*.db /tmp/st_$tag/*.db 2>/dev/null | head -1) > $O/stats_$tag.txt 2>&1; echo "== $tag"; grep "build_sci\|nonbonded" $O/stats_$tag.txt | cut -c1-110; }
ARGS="24 1 seq" st ala_alone GO_ITERS=1 GO_PHASES=1 REMD_OVERLAP=0
ARGS="16 1 seq dhfr" st dhfr_alone GO_STEPS=100 GO_ITERS=1 GO_PHASES=1 REMD_OVERLAP=0
timeout 900 python -m pytest tests/test_forcefield_parity.py -m gpu -q -x 2>&1 | tail -3 | tee $O/pytest.txt

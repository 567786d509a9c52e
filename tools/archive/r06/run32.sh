#!/bin/bash
# round 6, call 32: molecules ordered along the Hilbert curve on 2^b cells per edge (default) against the Z-order curve on 0.45 nm cells
export TMPDIR=/tmp
ROOT=$(pwd); O=$ROOT/gpurun_out/r06_32; mkdir -p $O
P="python tools/phase_probe.py"
{
for m in hilbert morton hilbert morton; do env GO_ITERS=4 GO_PHASES=2 REMD_NB_CURVE=$m $P 24 1 seq; done
for m in hilbert morton hilbert morton; do env GO_ITERS=6 GO_STEPS=100 GO_PHASES=2 REMD_NB_CURVE=$m $P 16 1 seq dhfr; done
for m in hilbert morton; do env GO_ITERS=3 GO_PHASES=1 REMD_NB_CURVE=$m $P 8 1 seq hostguest; done
} 2>&1 | grep -v "amdgpu.ids\|per-replica\|host enqueue" | cut -c1-260 | sed 's/ first .*//' | tee $O/probe.txt
st() { tag=$1; shift
  (cd /tmp && rm -rf /tmp/st_$tag && env "$@" rocprofv3 --kernel-trace -d /tmp/st_$tag -o kt -- python $ROOT/tools/phase_probe.py ${ARGS} > $O/run_$tag.txt 2>&1)
  python tools/rocpd_stats.py $(ls /tmp/st_$tag/*/*.db /tmp/st_$tag/*.db 2>/dev/null | head -1) > $O/stats_$tag.txt 2>&1; echo "== $tag"; grep "build_sci\|nonbonded\|gather_pos" $O/stats_$tag.txt | cut -c1-110; }
ARGS="24 1 seq" st ala_hilbert GO_ITERS=1 GO_PHASES=1 REMD_OVERLAP=0
ARGS="24 1 seq" st ala_morton GO_ITERS=1 GO_PHASES=1 REMD_OVERLAP=0 REMD_NB_CURVE=morton
ARGS="16 1 seq dhfr" st dhfr_hilbert GO_STEPS=100 GO_ITERS=1 GO_PHASES=1 REMD_OVERLAP=0
ARGS="16 1 seq dhfr" st dhfr_morton GO_STEPS=100 GO_ITERS=1 GO_PHASES=1 REMD_OVERLAP=0 REMD_NB_CURVE=morton
timeout 900 python -m pytest tests/test_forcefield_parity.py tests/test_phases_gpu.py -m gpu -q -x 2>&1 | tail -3 | tee $O/pytest.txt

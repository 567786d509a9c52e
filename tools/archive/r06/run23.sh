#!/bin/bash
# round 6, call 23: knock-out probes of the global atomics (results wrong on purpose): z inverse + gather without atomics / without the
# gather; forces.hip without add_force (pair i / j forces, listed terms)
export TMPDIR=/tmp
ROOT=$(pwd); O=$ROOT/gpurun_out/r06_23; mkdir -p $O
st() { tag=$1; shift
  (cd /tmp && rm -rf /tmp/st_$tag && env "$@" rocprofv3 --kernel-trace -d /tmp/st_$tag -o kt -- python $ROOT/tools/phase_probe.py ${ARGS} > $O/run_$tag.txt 2>&1)
  python tools/rocpd_stats.py $(ls /tmp/st_$tag/*/*.db /tmp/st_$tag/*.db 2>/dev/null | head -1) > $O/stats_$tag.txt 2>&1; echo "== $tag"; head -12 $O/stats_$tag.txt | cut -c1-110; }
for v in zi_noatom zi_nogather f_noatom; do
ARGS="24 1 seq" st ala_$v GO_ITERS=1 GO_PHASES=1 REMD_OVERLAP=0 AB_LIB=$ROOT/openmmtools_amd/libremd_hip_$v.so
ARGS="16 1 seq dhfr" st dhfr_$v GO_STEPS=100 GO_ITERS=1 GO_PHASES=1 REMD_OVERLAP=0 AB_LIB=$ROOT/openmmtools_amd/libremd_hip_$v.so
done

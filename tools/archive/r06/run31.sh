#!/bin/bash
# round 6, call 31: the ranking of the molecule sort on several workgroups per replica (rank_groups_kernel): same order, same digests
export TMPDIR=/tmp
ROOT=$(pwd); O=$ROOT/gpurun_out/r06_31; mkdir -p $O
{
AB_LIBS=",sortold" python tools/probes/cmp_list_builders2.py 16 100 3 1
AB_LIBS=",sortold" python tools/probes/cmp_list_builders2.py 16 100 3 2
P="python tools/phase_probe.py"
for lib in "" sortold "" sortold; do
  L=""; [ -n "$lib" ] && L="AB_LIB=$ROOT/openmmtools_amd/libremd_hip_$lib.so"
  env GO_ITERS=5 GO_STEPS=100 GO_PHASES=2 $L $P 16 1 seq dhfr
done
} 2>&1 | grep -v "amdgpu.ids\|per-replica\|host enqueue" | cut -c1-300 | sed 's/ first .*//' | tee $O/probe.txt

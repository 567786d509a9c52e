#!/bin/bash
# usage (GPU box, repo root): tools/mesh_variants.sh name1 name2 ...   ('base' = libremd_hip.so)
# stand-alone (REMD_OVERLAP=0) kernel durations of the mesh pipeline for variant builds, then ms per 500 steps with overlap
export TMPDIR=/tmp
ROOT=$(pwd)
for v in "$@"; do
  lib=$ROOT/openmmtools_amd/libremd_hip_$v.so; [ "$v" = base ] && lib=$ROOT/openmmtools_amd/libremd_hip.so
  (cd /tmp && rm -rf /tmp/mv_$v && AB_LIB=$lib REMD_OVERLAP=0 rocprofv3 --kernel-trace --stats -d /tmp/mv_$v -o kt -- python $ROOT/tools/launch_bound_check.py 24 > /dev/null 2>&1)
  echo "== $v stand-alone"; python tools/rocpd_stats.py $(find /tmp/mv_$v -name "*.db" | head -1) 2>/dev/null | grep -E "pme_|nonbonded_sci2|integrate_chain" | cut -c1-120
  echo "== $v overlapped"; AB_LIB=$lib timeout 300 python tools/launch_bound_check.py 24 2>&1 | tail -1
done

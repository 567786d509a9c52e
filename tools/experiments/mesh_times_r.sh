#!/bin/bash
# usage (GPU box): tools/mesh_times_r.sh variant R...   in-kernel phase stamps for several replica counts, stand-alone
v=$1; shift
lib=$PWD/openmmtools_amd/libremd_hip_$v.so; [ $v = base ] && lib=$PWD/openmmtools_amd/libremd_hip.so
for R in "$@"; do
  echo "== $v R=$R stand-alone"; AB_LIB=$lib REMD_MESH_TIMES=1 REMD_OVERLAP=0 timeout 300 python tools/launch_bound_check.py $R 2>&1 | grep -E "mesh\]|R $R|Error" | tail -6
done

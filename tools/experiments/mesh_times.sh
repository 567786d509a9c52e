#!/bin/bash
# usage (GPU box): tools/mesh_times.sh variant...   in-kernel phase stamps of pme_mesh_kernel (launch 200), stand-alone and overlapped
for v in "$@"; do
  lib=$PWD/openmmtools_amd/libremd_hip_$v.so; [ $v = base ] && lib=$PWD/openmmtools_amd/libremd_hip.so
  echo "== $v stand-alone"; AB_LIB=$lib REMD_MESH_TIMES=1 REMD_OVERLAP=0 timeout 300 python tools/launch_bound_check.py 24 2>&1 | grep -E "mesh\]|R 24|Error" | tail -6
  echo "== $v overlapped";  AB_LIB=$lib REMD_MESH_TIMES=1 timeout 300 python tools/launch_bound_check.py 24 2>&1 | grep -E "mesh\]|R 24|Error" | tail -6
done

#!/bin/bash
# usage: tools/build_variant.sh <name> <source.hip> [-DFLAG ...]   builds openmmtools_amd/libremd_hip_<name>.so: one translation unit
# recompiled with extra flags, the other objects of the regular build linked in (A/B runs: HipEngine(lib_path=...) / tools)
set -e
name=$1; src=$2; shift 2
cd "$(dirname "$0")/../openmmtools_amd/csrc"
make -s
base=${src%.hip}
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -Wno-unused-value -Wno-unused-result "$@" -c $src -o /tmp/${base}_$name.o
objs=""
for o in $(sed -n "s/^SRCS = //p" Makefile | sed "s/\.hip/.o/g"); do if [ "$o" = "$base.o" ]; then objs="$objs /tmp/${base}_$name.o"; else objs="$objs $o"; fi; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs -o ../libremd_hip_$name.so
echo built libremd_hip_$name.so

#!/bin/bash
# round 5, call 15: where the chain's first and last segments go (finer stamps), and the force-independent loads issued before the wait for the forces
export TMPDIR=/tmp
O=gpurun_out/r05_15; mkdir -p $O
for v in stamps earlystamps; do echo "== $v"; AB_LIB=$PWD/openmmtools_amd/libremd_hip_$v.so python tools/chain_segments.py 24 2>&1 | grep -v "^HIP\|^ROCm\|amdgpu.ids"; done | tee $O/segments.txt
python tools/ab_libs.py --R 24 --system alanine --rounds 3 tree early 2>&1 | grep -v "^HIP\|^ROCm\|amdgpu.ids" | tee $O/ab.txt
python tools/ab_libs.py --R 8 --system hostguest --rounds 2 tree early 2>&1 | grep -v "^HIP\|^ROCm\|amdgpu.ids" | tee -a $O/ab.txt

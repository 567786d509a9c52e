#!/bin/bash
# round 5, call 14: the chain's momentum sum as one exchange of tagged words instead of accumulate + arrive + read back
export TMPDIR=/tmp
O=gpurun_out/r05_14; mkdir -p $O
python tools/ab_libs.py --R 24 --system alanine --rounds 3 --no-insitu chainbase tree > /dev/null 2>&1   # (warm the box)
python tools/ab_libs.py --R 24 --system alanine --rounds 3 chainbase tree 2>&1 | grep -v "^HIP\|^ROCm\|amdgpu.ids" | tee $O/ab.txt
python tools/ab_libs.py --R 8 --system hostguest --rounds 2 chainbase tree 2>&1 | grep -v "^HIP\|^ROCm\|amdgpu.ids" | tee -a $O/ab.txt
timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider -x 2>&1 | grep -v "^HIP version\|^ROCm version\|^Hostname\|^Librccl" | tail -6 | tee $O/pytest_gpu.txt

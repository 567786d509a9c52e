#!/bin/bash
# round 5, call 10: free atoms four to an integrator unit (DHFR: 33 -> 32 chain workgroups per replica = two rounds of the chip instead of three)
export TMPDIR=/tmp
O=gpurun_out/r05_10; mkdir -p $O
python tools/split_sweep.py auto 16 dhfr 2>&1 | tail -1 | cut -c60-220 | tee $O/ab.txt
python tools/split_sweep.py auto 24 alanine 2>&1 | tail -1 | cut -c60-220 | tee -a $O/ab.txt
python tools/split_sweep.py auto 8 hostguest 2>&1 | tail -1 | cut -c60-220 | tee -a $O/ab.txt
python tools/bench_configs.py 5 2 2>/dev/null | cut -c1-200 | tee -a $O/ab.txt
timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | grep -v "^HIP version\|^ROCm version\|^Hostname\|^Librccl" | tail -6 | tee $O/pytest_gpu.txt

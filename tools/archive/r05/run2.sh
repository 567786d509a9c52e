#!/bin/bash
# round 5, call 2: pair lists one step behind + sorted positions written by the integrator chain: A/B against the exact lists, skin sweep, GPU suite
export TMPDIR=/tmp
O=gpurun_out/r05_2; mkdir -p $O
for i in 1 2; do
REMD_LIST_PIPE=0 python tools/split_sweep.py auto 24 alanine 2>&1 | tail -1 | cut -c60-200
python tools/split_sweep.py auto 24 alanine 2>&1 | tail -1 | cut -c60-200
done 2>&1 | tee $O/ab.txt
for skin in 0.03 0.045 0.08; do REMD_LIST_SKIN=$skin python tools/split_sweep.py auto 24 alanine 2>&1 | tail -1 | cut -c60-200; done 2>&1 | tee -a $O/ab.txt
REMD_LIST_PIPE=0 python tools/split_sweep.py auto 8 alanine 2>&1 | tail -1 | cut -c60-200 | tee -a $O/ab.txt
python tools/split_sweep.py auto 8 alanine 2>&1 | tail -1 | cut -c60-200 | tee -a $O/ab.txt
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 | tee $O/pytest_gpu.txt
python bench.py --no-cpu-baseline 2>/dev/null | head -c 400 | tee $O/bench.txt; echo

#!/bin/bash
# round 5, call 1: pair-kernel variants (deferred j atomics, dual steps, 3 / 4 waves per SIMD) A/B in one call + list statistics + GPU suite
export TMPDIR=/tmp
O=gpurun_out/r05_1; mkdir -p $O
timeout 900 python tools/ab_libs.py --rounds 2 head r5base jdefer dual4 tree dual3ns dualonly3 noj > $O/ab_alanine.txt 2>&1
cat $O/ab_alanine.txt
REMD_DEBUG=1 timeout 120 python tools/sci_microbench.py "" 2>&1 | grep -i "sci list\|nonbonded" | head -5 | tee $O/list_stats.txt
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 | tee $O/pytest_gpu.txt

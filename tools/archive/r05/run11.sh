#!/bin/bash
# round 5, call 11: G concurrent handles of R/G replicas against one handle of R
export TMPDIR=/tmp
O=gpurun_out/r05_11; mkdir -p $O
python tools/group_overlap.py 24 alanine 1 2 3 4 1 2>&1 | grep -v "^HIP\|^ROCm" | tee $O/groups.txt
python tools/group_overlap.py 16 dhfr 1 2 2>&1 | grep -v "^HIP\|^ROCm" | tee -a $O/groups.txt
python tools/group_overlap.py 8 hostguest 1 2 2>&1 | grep -v "^HIP\|^ROCm" | tee -a $O/groups.txt

#!/bin/bash
# round 5, call 8: explicit DPP all-reduce of the j forces A/B (tree vs -DSCI_DPP_ASM=0), pair-kernel PMC attribution, GPU suite
export TMPDIR=/tmp
O=gpurun_out/r05_8; mkdir -p $O
timeout 900 python tools/ab_libs.py --rounds 3 nodpp tree 2>&1 | grep -v amdgpu.ids | tee $O/ab_dpp.txt
bash tools/pmc_pair_attrib.sh $O/pmc_pair_attrib.md > /dev/null 2>&1; cat $O/pmc_pair_attrib.md
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | grep -v "^HIP version\|^ROCm version\|^Hostname\|^Librccl" | tail -6 | tee $O/pytest_gpu.txt

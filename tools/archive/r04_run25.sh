#!/bin/bash
# DHFR (config 5 shape: 16 replicas) kernel table at the end of round 4
export TMPDIR=/tmp
mkdir -p gpurun_out/r04_z
DHFR_STEPS=100 DHFR_ITERS=4 python tools/dhfr_profile.py dhfr 16 > gpurun_out/r04_z/dhfr_plain.txt 2>&1
REMD_PROFILE=1 DHFR_STEPS=100 DHFR_ITERS=3 python tools/dhfr_profile.py dhfr 16 > gpurun_out/r04_z/dhfr_scopes.txt 2>&1
cd /tmp && DHFR_STEPS=100 DHFR_ITERS=3 rocprofv3 --kernel-trace --stats -d /tmp/prof_dhfr -o dhfr -- python $GRAFT_REPO_ROOT/tools/dhfr_profile.py dhfr 16 > $GRAFT_REPO_ROOT/gpurun_out/r04_z/dhfr_prof.txt 2>&1
cd $GRAFT_REPO_ROOT
python tools/rocpd_stats.py $(find /tmp/prof_dhfr -name "*.db" | head -1) gpurun_out/r04_z/dhfr_kernel_stats.md > /dev/null
head -22 gpurun_out/r04_z/dhfr_kernel_stats.md | cut -c1-130
cat gpurun_out/r04_z/dhfr_plain.txt | tail -5

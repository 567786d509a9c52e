#!/bin/bash
# round 5, call 7: pair work items dispatched longest list first (REMD_NB_RANK) A/B, stand-alone and in situ, three systems + bit-identity tests
export TMPDIR=/tmp
O=gpurun_out/r05_7; mkdir -p $O
for v in 0 1 0 1; do REMD_NB_RANK=$v python tools/split_sweep.py auto 24 alanine standalone 2>&1 | tail -1 | cut -c60-330; done 2>&1 | tee $O/ab.txt
for sysR in "alanine 24" "alanine 8" "hostguest 8" "dhfr 16"; do set -- $sysR
  for v in 0 1 0 1; do REMD_NB_RANK=$v python tools/split_sweep.py auto $2 $1 2>&1 | tail -1 | cut -c60-220; done
done 2>&1 | tee -a $O/ab.txt
timeout 900 python -m pytest tests/test_forcefield_parity.py tests/test_openmm_fixture.py tests/test_compat_groups.py tests/test_distributed_gpu.py -m gpu -x -q 2>&1 | tail -4 | tee $O/pytest.txt

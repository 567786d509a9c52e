#!/bin/bash
# round 5, call 12: the chain compiled for two workgroups per CU on DHFR (one round of the chip instead of two) + instruction-cache counters of the headline step
export TMPDIR=/tmp
O=gpurun_out/r05_12; mkdir -p $O
for v in 0 1 0 1; do REMD_CHAIN_TWO=$v python tools/split_sweep.py auto 16 dhfr 2>&1 | tail -1 | cut -c40-220 | tee -a $O/ab.txt; done
REMD_CHAIN_TWO=1 python tools/split_sweep.py auto 24 alanine 2>&1 | tail -1 | cut -c40-220 | tee -a $O/ab.txt
python tools/split_sweep.py auto 24 alanine 2>&1 | tail -1 | cut -c40-220 | tee -a $O/ab.txt
timeout 600 python -m pytest tests/test_forcefield_parity.py -m gpu -q -p no:cacheprovider -k "dhfr or DHFR" 2>&1 | tail -2 | tee -a $O/ab.txt
timeout 900 tools/pmc_icache.sh $O/pmc_icache.md 2>&1 | cut -c1-330

#!/bin/bash
# integrator chain at a register cap for grids larger than the chip (DHFR): REMD_CHAIN_WAVES = 0 (320 VGPRs) / 2 (256) / 3 (168)
export TMPDIR=/tmp
for w in 0 2 3; do
echo "== REMD_CHAIN_WAVES=$w"
REMD_CHAIN_WAVES=$w DHFR_STEPS=100 DHFR_ITERS=4 python tools/dhfr_profile.py dhfr 16 2>&1 | tail -3 | tr '\n' ' '; echo
done
timeout 600 python -m pytest tests/test_forcefield_parity.py tests/test_harmonic_parity.py tests/test_work_parity.py -m gpu -x -q -k "config5 or dhfr or chain or langevin or splitting or harmonic" 2>&1 | tail -2

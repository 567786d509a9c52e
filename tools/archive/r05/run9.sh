#!/bin/bash
# round 5, call 9: Coulomb force table as 128 bins per octave x quadratic (tree) against 32 x cubic (-DCTAB_QUAD=0): stand-alone + in situ, three systems; parity tests
export TMPDIR=/tmp
O=gpurun_out/r05_9; mkdir -p $O
timeout 900 python tools/ab_libs.py --rounds 3 cubic tree 2>&1 | grep -v amdgpu.ids | tee $O/ab_table.txt
timeout 600 python tools/ab_libs.py --rounds 2 --system hostguest --R 8 cubic tree 2>&1 | grep -v amdgpu.ids | tee -a $O/ab_table.txt
timeout 900 python tools/ab_libs.py --rounds 2 --system dhfr --R 16 --steps 100 cubic tree 2>&1 | grep -v amdgpu.ids | tee -a $O/ab_table.txt
timeout 900 python -m pytest tests/test_forcefield_parity.py tests/test_openmm_fixture.py tests/test_coulomb_table.py tests/test_harmonic_parity.py -m gpu -q -p no:cacheprovider 2>&1 | grep -v "^HIP version\|^ROCm version\|^Hostname\|^Librccl" | tail -5 | tee $O/pytest.txt

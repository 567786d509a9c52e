#!/bin/bash
# round 5, call 6: swap-all as a rendezvous of the replica slots: parity (all kernels) + ms per call at R = 64 / 128 / 192 / 256 in both acceptance regimes
export TMPDIR=/tmp
O=gpurun_out/r05_6; mkdir -p $O
timeout 900 python -m pytest tests/test_mix_parity.py tests/test_reference_golden.py -m gpu -x -q 2>&1 | tail -12 | tee $O/pytest_mix.txt
for cfg in "REMD_MIX_RDV=0" "REMD_MIX_RDV=1 MIX_SET_BETA=1" "REMD_MIX_RDV=1 MIX_SET_BETA=1 REMD_MIX_RDV_PT=0" "REMD_MIX_RDV=0 MIX_MATRIX=hot" "REMD_MIX_RDV=1 MIX_MATRIX=hot"; do
  echo "== $cfg"; env $cfg REMD_MIX_DEBUG=1 timeout 300 python tools/mix_microbench.py 64 128 192 256 2>&1 | grep -v amdgpu.ids | grep "^R \|mix-rdv" | awk '/mix-rdv/{c++; if (c%7==1) print; next} {print}'
done 2>&1 | tee $O/mix_microbench.txt

"""Stand-alone run of the matrix-core XY pass (csrc/dft_mfma.hip) through the test hook: 888 planes of 75 x 75 (the
24-replica alanine mesh), for rocprofv3 --kernel-trace / --pmc.  usage: python tools/dft_mfma_bench.py [planes] [n] [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from openmmtools_amd._engine import HipEngine
p = int(sys.argv[1]) if len(sys.argv) > 1 else 888
n = int(sys.argv[2]) if len(sys.argv) > 2 else 75
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
rng = np.random.default_rng(0)
a = (rng.normal(size=(p, n, n)) + 1j * rng.normal(size=(p, n, n))).astype(np.complex64)
eng = HipEngine()
for _ in range(reps):
    b = eng.test_xy_mfma(a, mode=0)
print('round trip err', np.abs(b / (n * n) - a).max())

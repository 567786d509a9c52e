"""GPU microbenchmark of the swap-all kernel (ns / attempt) at the BASELINE replica counts."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from openmmtools_amd._engine import HipEngine
eng = HipEngine(); eng.seed(0xC0FFEE)
rng = np.random.default_rng(0)
for R in (4, 16, 24, 64, 128):
    u = np.outer(rng.normal(scale=3.0, size=R), np.linspace(0.5, 1.5, R)) + rng.normal(scale=0.5, size=(R, R))
    labels = np.arange(R)
    eng.mix_host('swap-all', 0, u, labels)
    eng.profile_enable(2); eng.profile_reset()
    for it in range(5):
        eng.mix_host('swap-all', it, u, labels)
    n, ms = eng.profile_get('mix_swap_all')
    print('R', R, 'attempts', R ** 3, 'ms/call', ms / n, 'ns/attempt', 1e6 * ms / n / R ** 3)

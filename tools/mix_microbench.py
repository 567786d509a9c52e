"""GPU microbenchmark of the swap-all kernel (ns / attempt) at the BASELINE replica counts."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from openmmtools_amd._engine import HipEngine
eng = HipEngine(lib_path=os.environ.get("AB_LIB") or None); eng.seed(0xC0FFEE)
rng = np.random.default_rng(0)
kind = os.environ.get('MIX_MATRIX', 'pt')      # 'pt': parallel-tempering-like (few % acceptance); 'hot': ~50 % acceptance
# MIX_SET_BETA=1: the states' beta on the handle (remd_set_states), as a sampler has them: the rendezvous kernel may then take the
# separable shortcut on the 'pt' matrix (REMD_MIX_RDV / REMD_MIX_RDV_PT / REMD_MIX_FLOW / REMD_MIX_PRE pin the kernels)
Rs = [int(a) for a in sys.argv[1:]] or [4, 16, 24, 48, 64, 96, 128, 192]
for R in Rs:
    if kind == 'hot':
        u = np.outer(rng.normal(scale=3.0, size=R), np.linspace(0.5, 1.5, R)) + rng.normal(scale=0.5, size=(R, R))
    else:   # 4500 degrees of freedom, temperatures 300..600 K geometric: U_k ~ equipartition mean + fluctuation
        T = np.geomspace(300.0, 600.0, R); kT = 0.0083145 * T
        u = np.outer(-30000.0 + 0.5 * kT * 4500 + rng.normal(size=R) * np.sqrt(2250.0) * kT, 1.0 / kT)
        if os.environ.get('MIX_SET_BETA'):
            eng.set_states(1.0 / kT)
    labels = np.arange(R)
    eng.mix_host('swap-all', 0, u, labels)
    eng.profile_enable(2); eng.profile_reset()
    for it in range(5):
        eng.mix_host('swap-all', it, u, labels)
    n, ms = eng.profile_get('mix_swap_all')
    _, nacc, nprop, _ = eng.mix_host('swap-all', 99, u, labels)
    print('R', R, 'attempts', R ** 3, 'ms/call', ms / n, 'ns/attempt', 1e6 * ms / n / R ** 3, 'acceptance', float(nacc.sum()) / max(1.0, float(nprop.sum())))

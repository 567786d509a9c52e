"""Is the integrator chain's duration set by the X-H (SHAKE) wave?  Times the chain launches with and without the solute's
SHAKE clusters (the hydrogens become free atoms: wrong physics, same kernel otherwise)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from openmmtools_amd import testsystems as ts
from openmmtools_amd.system import system_to_desc
from openmmtools_amd._engine import HipEngine
KB = 0.008314462618153242
al = ts.AlanineDipeptideExplicit()
box = np.diag(al.system.getDefaultPeriodicBoxVectors())
R = 24
for drop in (False, True):
    d = system_to_desc(al.system)
    if drop:
        d['shake_atoms'] = np.zeros((0, 4), np.int32); d['shake_dist'] = np.zeros((0, 3))
    eng = HipEngine()
    eng.set_system(d); eng.set_states(np.full(R, 1 / (KB * 300.0)))
    eng.set_integrator('V R R O R R V', 0.0005 if drop else 0.002, 1.0, 200, True, 1e-8)
    eng.set_replicas(R, 0, np.tile(al.positions, (R, 1, 1)), None, np.tile(box, (R, 1)), np.arange(R))
    eng.propagate(0)
    eng.profile_enable(2); eng.profile_reset()
    eng.propagate(1)
    n, ms = eng.profile_get('integrate_chain')
    print('shake clusters dropped' if drop else 'with shake clusters  ', 'chain launches', n, 'avg us', 1e3 * ms / n)
    eng.profile_enable(0); eng.close()

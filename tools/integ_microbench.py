"""GPU microbenchmark of the integrator chain kernel per token (diagnostic)."""
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
desc = system_to_desc(al.system)
if len(sys.argv) > 1 and sys.argv[1] == 'nocmm':
    desc['cmm_frequency'] = 0
eng = HipEngine()
eng.set_system(desc); eng.set_states(np.full(R, 1 / (KB * 300.0)))
eng.set_integrator('V R R O R R V', 0.002, 1.0, 1, True, 1e-8)
eng.set_replicas(R, 0, np.tile(al.positions, (R, 1, 1)), None, np.tile(box, (R, 1)), np.arange(R))
eng.propagate(0)
eng.profile_enable(2)
for toks in ('V', 'R', 'O', 'RR', 'RROR', 'VRRORR'):
    eng.get_forces()
    eng.profile_reset()
    for k in range(10):
        eng.step(toks, iteration=1, first_step=k)
    n, ms = eng.profile_get('integrate_chain')
    print(toks, 'launches', n, 'avg us', 1e3 * ms / max(n, 1))

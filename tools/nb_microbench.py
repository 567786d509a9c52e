"""GPU microbenchmark of the direct-space nonbonded kernel: time per launch vs cutoff / sorting (diagnostic)."""
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
for cutoff in (0.2, 0.6, 1.0):
    desc = system_to_desc(al.system)
    desc['cutoff'] = cutoff
    desc['switch_distance'] = cutoff * 0.85
    eng = HipEngine()
    eng.set_system(desc); eng.set_states(np.full(R, 1 / (KB * 300.0)))
    eng.set_integrator('V R O R V', 0.001, 1.0, 1, True, 1e-8)
    eng.set_replicas(R, 0, np.tile(al.positions, (R, 1, 1)), None, np.tile(box, (R, 1)), np.arange(R))
    eng.get_forces()
    eng.profile_enable(2); eng.profile_reset()
    for _ in range(10):
        eng.lib.remd_get_forces  # noqa
        eng.get_forces()
    out = {k: eng.profile_get(k) for k in ('nonbonded', 'nb_gather', 'nb_sort', 'pme_fft', 'pme_bin', 'pme_gather')}
    print('cutoff', cutoff, {k: round(1e3 * v[1] / max(1, v[0]), 1) for k, v in out.items()})
    eng.close()

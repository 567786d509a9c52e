"""Stand-alone durations of the mesh launches (spread + z forward | XY plane pass | z inverse + gather) and of the pair kernel on
the headline system (or DHFR): force evaluations only, stream overlap off, per-scope events.
usage: [AB_LIB=...] python tools/mesh_standalone.py [R] [dhfr]"""
import os, sys
os.environ.setdefault('REMD_OVERLAP', '0')
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from openmmtools_amd import testsystems as ts
from openmmtools_amd.system import system_to_desc
from openmmtools_amd._engine import HipEngine
KB = 0.008314462618153242
al = ts.DHFRExplicit() if (len(sys.argv) > 2 and sys.argv[2] == 'dhfr') else ts.AlanineDipeptideExplicit()
box = np.diag(al.system.getDefaultPeriodicBoxVectors())
R = int(sys.argv[1]) if len(sys.argv) > 1 else 24
eng = HipEngine(lib_path=os.environ.get('AB_LIB') or None)
eng.set_system(system_to_desc(al.system)); eng.set_states(np.full(R, 1 / (KB * 300.0)))
eng.set_integrator('V R O R V', 0.001, 1.0, 1, True, 1e-8)
eng.set_replicas(R, 0, np.tile(al.positions, (R, 1, 1)), None, np.tile(box, (R, 1)), np.arange(R))
eng.get_forces()
eng.profile_enable(2); eng.profile_reset()
for _ in range(20):
    eng.get_forces()
out = {k: eng.profile_get(k) for k in ('pme_fft', 'pme_xy', 'pme_zinv_gather', 'pme_bin', 'nonbonded', 'nb_gather')}
us = {k: round(1e3 * v[1] / max(1, v[0]), 1) for k, v in out.items()}
us['pme_spread_zfwd'] = round(us['pme_fft'] - us['pme_xy'] - us['pme_zinv_gather'], 1)
print(os.environ.get('AB_LIB', 'base').split('/')[-1], 'R', R, 'us per launch', us)

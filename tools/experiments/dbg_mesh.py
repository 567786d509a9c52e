import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from openmmtools_amd import testsystems as ts
from openmmtools_amd.system import system_to_desc
from openmmtools_amd._engine import HipEngine
KB = 0.008314462618153242
al = ts.AlanineDipeptideExplicit()
box = np.diag(al.system.getDefaultPeriodicBoxVectors())
R = int(sys.argv[1]) if len(sys.argv) > 1 else 3
res = {}
for flag in ("1",):
    os.environ['REMD_PME_MESH1'] = flag
    eng = HipEngine(lib_path=os.environ.get('AB_LIB') or None)
    eng.set_system(system_to_desc(al.system)); eng.set_states(np.full(R, 1 / (KB * 300.0)))
    eng.set_integrator('V R R O R R V', 0.002, 1.0, 20, True, 1e-8)
    eng.set_replicas(R, 0, np.tile(al.positions, (R, 1, 1)), None, np.tile(box, (R, 1)), np.arange(R))
    for k in range(3):
        rows, U = eng.compute_energies(want_potential=True)
        c = eng.energy_components()
        f = eng.get_forces()
        print('mesh1', flag, 'eval', k, 'U', U[:3], 'recip', [ci['pme_reciprocal'] for ci in c][:3], 'fmax', np.abs(f).max())
    res[flag] = (U, f)
    eng.close()
print('force diff', np.abs(res['0'][1] - res['1'][1]).max())

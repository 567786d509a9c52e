"""resident small-molecule kernel against the regular launches: size of the difference after n steps"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from openmmtools_amd import testsystems as ts
from openmmtools_amd.system import system_to_desc
from openmmtools_amd._engine import HipEngine
KB = 0.008314462618153242
al = ts.AlanineDipeptideVacuum()
desc = system_to_desc(al.system)
for splitting in ('V R V', 'V R O R V', 'V R R O R R V'):
    for n in (1, 2, 10, 100):
        out = []
        for flag in ('1', '0'):
            os.environ['REMD_RESIDENT'] = flag
            eng = HipEngine()
            eng.set_system(desc); eng.set_states(1.0 / (KB * np.array([300.0, 450.0])))
            eng.set_integrator(splitting, 0.002, 1.0, n, True, 1e-8); eng.seed(11)
            eng.set_replicas(2, 0, np.tile(al.positions, (2, 1, 1)), None, np.zeros((2, 3)), np.arange(2))
            eng.propagate(0)
            x, v = eng.get_replicas()[:2]
            out.append((x.copy(), v.copy())); eng.close()
        print(splitting, n, 'max |dx| %.3e  max |dv| %.3e   moved %.3e' % (np.abs(out[0][0] - out[1][0]).max(), np.abs(out[0][1] - out[1][1]).max(), np.abs(out[0][0] - al.positions).max()), flush=True)

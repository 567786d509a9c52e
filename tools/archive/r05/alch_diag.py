import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), 'tests'))
import numpy as np
import test_alchemy_expressions as T
from openmmtools_amd._engine import HipEngine
from openmmtools_amd.system import system_to_desc
for exception in (False, True):
    for (sigma, epsilon), by_r in T._groups().items():
        rs = sorted(by_r)
        system = T._two_particles(sigma, epsilon, exception)
        x = T._positions(rs)
        eng = HipEngine()
        eng.set_system(system_to_desc(system))
        eng.set_states(np.full(4, 1.0 / (T.KB * 300.0)), np.array(T.LAMBDAS), np.ones(4), None)
        eng.set_integrator('V R O R V', 0.001, 1.0, 1, True, 1e-8)
        eng.set_replicas(len(rs), 0, x, None, np.tile([T.L] * 3, (len(rs), 1)), np.zeros(len(rs), dtype=int))
        rows = np.asarray(eng.compute_energies()) * (T.KB * 300.0)
        eng.close()
        want = np.array([[by_r[r][lam][1 if exception else 0] for lam in T.LAMBDAS] for r in rs])
        np.set_printoptions(precision=6, linewidth=200)
        print('exception', exception, 'sigma %.3f eps %.3f' % (sigma, epsilon), 'r', np.round(rs, 3))
        print(' want', want.tolist()); print(' diff', (rows - want).tolist())

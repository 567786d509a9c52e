import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from openmmtools_amd import testsystems as ts, alchemy
from openmmtools_amd.system import system_to_desc
from openmmtools_amd._engine import HipEngine
from oracle.forcefield import ForceFieldOracle
KB = 0.008314462618153242
hg = ts.HostGuestExplicit()
region = alchemy.AlchemicalRegion(alchemical_atoms=range(126, 156))
system = alchemy.AbsoluteAlchemicalFactory().create_alchemical_system(hg.system, region)
desc = system_to_desc(system)
ff = ForceFieldOracle(desc)
box = np.diag(system.getDefaultPeriodicBoxVectors())
for (le, ls) in [(1.0, 1.0), (0.5, 1.0), (0.0, 1.0), (1.0, 0.2), (0.0, 0.2)]:
    eng = HipEngine()
    eng.set_system(desc)
    eng.set_states(np.array([1 / (KB * 300.0)]), np.array([ls]), np.array([le]), None)
    eng.set_integrator('V R R O R R V', 0.002, 1.0, 5, True, 1e-8)
    eng.set_replicas(1, 0, hg.positions[None], None, box[None], np.zeros(1, int))
    f = eng.get_forces()[0]
    x = eng.get_replicas()[0][0]
    e_ref, f_ref = ff.energy_forces(x, box, lambda_sterics=ls, lambda_electrostatics=le)
    err = np.sqrt(((f - f_ref) ** 2).sum(1))
    w = np.argsort(-err)[:6]
    print('le', le, 'ls', ls, 'rmse', np.sqrt((err ** 2).mean()), 'worst', w, err[w].round(2), 'alch worst', err[126:156].max().round(3), 'U', eng.compute_energies(want_potential=True)[1][0], e_ref)
    eng.close()

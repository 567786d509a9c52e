import sys, numpy as np
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
from openmmtools_amd import testsystems as ts, alchemy
from openmmtools_amd.system import system_to_desc
from openmmtools_amd._engine import HipEngine
from oracle.forcefield import ForceFieldOracle
KB = 0.008314462618153242
lj = ts.LennardJonesFluid(nparticles=216)
system = alchemy.AbsoluteAlchemicalFactory().create_alchemical_system(lj.system, alchemy.AlchemicalRegion(alchemical_atoms=range(10)))
lam = np.array([1.0, 0.6, 0.3, 0.0])
eng = HipEngine()
desc = system_to_desc(system)
eng.set_system(desc)
eng.set_states(np.full(4, 1.0 / (KB * 120.0)), lam, None, np.zeros(4))
eng.set_integrator('V R O R V', 0.002, 1.0, 5, True, 1e-8)
eng.seed(1)
x = np.tile(lj.positions, (4, 1, 1)); box = np.tile(np.diag(system.getDefaultPeriodicBoxVectors()), (4, 1))
eng.set_replicas(4, 0, x, None, box, np.arange(4))
rows, U = eng.compute_energies(want_potential=True)
ff = ForceFieldOracle(desc)
full = [ff.energy_forces(x[r], box[r], lambda_sterics=lam[r], forces=False)[0] for r in range(4)]
print('device potential', U)
print('oracle full     ', np.array(full))
print('rows diag kT    ', np.diag(rows) * KB * 120.0)

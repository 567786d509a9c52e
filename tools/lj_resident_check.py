"""ms per 500 MD steps of 16 x LennardJonesFluid(512) with alchemical atoms (BASELINE config 2): resident kernel timing experiments."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from openmmtools_amd import testsystems as ts, alchemy
from openmmtools_amd.system import system_to_desc
from openmmtools_amd._engine import HipEngine
KB = 0.008314462618153242
lj = ts.LennardJonesFluid(nparticles=512)
region = alchemy.AlchemicalRegion(alchemical_atoms=range(10))
system = alchemy.AbsoluteAlchemicalFactory().create_alchemical_system(lj.system, region)
box = np.diag(system.getDefaultPeriodicBoxVectors())
R = 16
lam = np.linspace(1.0, 0.0, R)
eng = HipEngine()
desc = system_to_desc(system)
print('cmm_frequency', desc['cmm_frequency'])
eng.set_system(desc); eng.set_states(np.full(R, 1 / (KB * 300.0)), lam, None, None)
eng.set_integrator('V R O R V', 0.001, 1.0, 500, True, 1e-8)
eng.set_replicas(R, 0, np.tile(lj.positions, (R, 1, 1)), None, np.tile(box, (R, 1)), np.arange(R))
eng.propagate(0)
for it in range(1, 4):
    t0 = time.perf_counter(); eng.propagate(it); t1 = time.perf_counter()
    print('wall ms', 1e3 * (t1 - t0), 'gpu ms', eng.last_timing()['propagate_ms'])

import cProfile, pstats, sys, os, io
sys.path.insert(0, os.getcwd())
import numpy as np
from openmmtools_amd import testsystems, states, mcmc, unit, alchemy
from openmmtools_amd.multistate import ReplicaExchangeSampler
from openmmtools_amd._engine import HipEngine
sys.path.insert(0, os.path.join(os.getcwd(), 'tools'))
import bench_configs as bc
lj = testsystems.LennardJonesFluid(nparticles=512)
ths = bc.alchemical_states(lj.system, range(10), np.ones(16), np.linspace(1.0, 0.0, 16))
move = mcmc.LangevinSplittingDynamicsMove(timestep=1.0 * unit.femtosecond, collision_rate=1.0 / unit.picosecond, n_steps=500, reassign_velocities=True, splitting='V R O R V')
s = ReplicaExchangeSampler(mcmc_moves=move, number_of_iterations=10 ** 9, engine=HipEngine(), seed=1)
s.create(ths, [states.SamplerState(lj.positions, box_vectors=lj.system.getDefaultPeriodicBoxVectors())])
s.run(3)
pr = cProfile.Profile(); pr.enable(); s.run(50); pr.disable()
st = io.StringIO(); pstats.Stats(pr, stream=st).sort_stats('tottime').print_stats(18); print(st.getvalue()[:3500])

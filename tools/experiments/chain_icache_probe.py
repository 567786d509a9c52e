"""Is the integrator chain's own time instruction fetch?  The same chain launched back to back (a splitting without V needs no force
evaluation between two MD steps, so nothing evicts the chain's code from the instruction cache) against the chain in the normal flow.
usage (under rocprofv3 --kernel-trace --stats --output-format csv): python chain_icache_probe.py "R R O R R" | "V R R O R R V"
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from openmmtools_amd import testsystems as ts
from openmmtools_amd.system import system_to_desc
from openmmtools_amd._engine import HipEngine
KB = 0.008314462618153242
al = ts.AlanineDipeptideExplicit()
box = np.diag(al.system.getDefaultPeriodicBoxVectors())
R = 24
eng = HipEngine()
eng.set_system(system_to_desc(al.system)); eng.set_states(np.full(R, 1 / (KB * 300.0)))
eng.set_integrator(sys.argv[1], 0.0005, 1.0, 200, True, 1e-8)
eng.set_replicas(R, 0, np.tile(al.positions, (R, 1, 1)), None, np.tile(box, (R, 1)), np.arange(R))
eng.propagate(0)
eng.propagate(1)
print('ok', sys.argv[1], np.isfinite(eng.get_replicas()[0]).all())

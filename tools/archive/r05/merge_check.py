"""The momentum sum inside one chain launch (REMD_CHAIN_MERGE=1) against two launches (=0): same integer sum, so the same trajectory bit
for bit -- at the headline size (24 replicas, 72 chain workgroups) and over 2500 steps, per library.  usage: merge_check.py lib ..."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from openmmtools_amd import testsystems as ts
from openmmtools_amd.system import system_to_desc
from openmmtools_amd._engine import HipEngine
KB = 0.008314462618153242
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
al = ts.AlanineDipeptideExplicit()
box = np.diag(al.system.getDefaultPeriodicBoxVectors())
d = system_to_desc(al.system, ewald_split='auto')
R = 24
x0 = np.tile(al.positions, (R, 1, 1)) + np.random.default_rng(7).normal(0, 0.002, (R,) + al.positions.shape)
ref = {}
for name in sys.argv[1:]:
    lib = None if name == 'tree' else os.path.join(root, 'openmmtools_amd', 'libremd_hip_%s.so' % name)
    for merge in ('0', '1', '0', '1'):
        os.environ['REMD_CHAIN_MERGE'] = merge
        eng = HipEngine(lib_path=lib, ewald_split='auto')
        eng.set_system(d); eng.set_states(1 / (KB * np.geomspace(300.0, 600.0, R)))
        eng.set_integrator('V R R O R R V', 0.002, 1.0, 500, True, 1e-8)
        eng.set_replicas(R, 0, x0, None, np.tile(box, (R, 1)), np.arange(R))
        sums = []
        for it in range(5):
            eng.propagate(it); sums.append(float(np.abs(eng.get_replicas()[0]).sum()))
        eng.close()
        ref.setdefault('first', sums)
        print('%-10s merge %s  |x| sums after each 500 steps: %s   equal to the first run up to iteration %d' % (
            name, merge, ' '.join('%.6f' % v for v in sums), sum(1 for a, b in zip(sums, ref['first']) if a == b)), flush=True)

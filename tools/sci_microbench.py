"""GPU microbenchmark of the direct-space pair kernels on a fixed configuration (24 x alanine dipeptide in water):
time per force evaluation of the 'nonbonded' / 'nonbonded_lj' profile classes for a list of environment variants.
usage: python tools/sci_microbench.py "" "REMD_NB_TILES=1" "REMD_NB_PERSIST_GRID=576" ...  (diagnostic)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from openmmtools_amd import testsystems as ts
from openmmtools_amd.system import system_to_desc
from openmmtools_amd._engine import HipEngine
KB = 0.008314462618153242
al = ts.AlanineDipeptideExplicit()
box = np.diag(al.system.getDefaultPeriodicBoxVectors())
R = 24
ref = None
for variant in sys.argv[1:] or ['']:
    keys = []
    for kv in variant.split():
        k, v = kv.split('=')
        os.environ[k] = v; keys.append(k)
    eng = HipEngine()
    eng.set_system(system_to_desc(al.system)); eng.set_states(np.full(R, 1 / (KB * 300.0)))
    eng.set_integrator('V R O R V', 0.001, 1.0, 1, True, 1e-8)
    eng.set_replicas(R, 0, np.tile(al.positions, (R, 1, 1)), None, np.tile(box, (R, 1)), np.arange(R))
    f = eng.get_forces()
    if ref is None: ref = f
    eng.profile_enable(2); eng.profile_reset()
    for _ in range(20):
        eng.get_forces()
    out = {k: eng.profile_get(k) for k in ('nonbonded', 'nonbonded_lj', 'nb_gather')}
    print('%-44s' % variant, {k: round(1e3 * v[1] / max(1, v[0]), 1) for k, v in out.items()},
          'max |dF| vs first variant %.3g' % np.abs(f - ref).max(), flush=True)
    eng.close()
    for k in keys: os.environ.pop(k)

import os, sys, numpy as np
sys.path.insert(0, os.getcwd())
from openmmtools_amd import testsystems as ts
from openmmtools_amd.system import system_to_desc
from openmmtools_amd._engine import HipEngine
KB = 0.008314462618153242
al = ts.DHFRExplicit()
box = np.diag(al.system.getDefaultPeriodicBoxVectors())
d = system_to_desc(al.system, ewald_split='auto')
out = {}
for name in ('', 'list1', 'listold'):
    lib = os.path.join(os.getcwd(), 'openmmtools_amd', 'libremd_hip_%s.so' % name) if name else None
    e = HipEngine(ewald_split='auto', lib_path=lib)
    e.set_system(d); e.set_states(np.full(2, 1 / (KB * 300.0)))
    e.set_integrator('V R R O R R V', 0.002, 1.0, 10, True, 1e-8)
    e.set_replicas(2, 0, np.tile(al.positions, (2, 1, 1)), None, np.tile(box, (2, 1)), np.arange(2))
    f = e.get_forces(); U = e.compute_energies(want_potential=True)[1]
    e.seed(5); e.propagate(0); x = e.get_replicas()[0]
    out[name] = (f.copy(), U.copy(), x.copy())
    print(name or 'new', 'U', U)
f0, U0, x0 = out['listold']
for name in ('', 'list1'):
    f, U, x = out[name]
    df = np.abs(f - f0)
    print(name or 'new', 'max |df|', df.max(), 'atoms differing', (df.max(axis=2) > 0).sum(), 'dU', U - U0, 'max dx', np.abs(x - x0).max())
    if df.max() > 0:
        r, a = np.unravel_index(df.max(axis=2).argmax(), df.shape[:2]); print('  worst atom', r, a, f[r, a], f0[r, a])

"""per-iteration digests of DHFR x R under two builds of the library (AB_LIBS: comma separated names, '' = the regular build)"""
import os, sys, hashlib, numpy as np
sys.path.insert(0, os.getcwd())
from openmmtools_amd import testsystems as ts
from openmmtools_amd.system import system_to_desc
from openmmtools_amd._engine import HipEngine
KB = 0.008314462618153242
R = int(sys.argv[1]); steps = int(sys.argv[2]); iters = int(sys.argv[3]); phases = int(sys.argv[4])
al = ts.DHFRExplicit()
box = np.diag(al.system.getDefaultPeriodicBoxVectors())
d = system_to_desc(al.system, ewald_split='auto')
for name in os.environ.get('AB_LIBS', ',listold').split(','):
    lib = os.path.join(os.getcwd(), 'openmmtools_amd', 'libremd_hip_%s.so' % name) if name else None
    e = HipEngine(ewald_split='auto', lib_path=lib)
    e.set_system(d); e.set_states(1 / (KB * np.geomspace(300.0, 600.0, R)))
    e.set_integrator('V R R O R R V', 0.002, 1.0, steps, True, 1e-8)
    e.seed(11)
    e.set_replicas(R, 0, np.tile(al.positions, (R, 1, 1)), None, np.tile(box, (R, 1)), np.arange(R))
    e.set_phases(phases)
    dig = []
    for it in range(iters):
        e.propagate(it)
        x = e.get_replicas()[0]
        dig.append(hashlib.sha1(x.tobytes()).hexdigest()[:8])
    print('%-8s R %d steps %d phases %d:' % (name or 'new', R, steps, phases), ' '.join(dig), flush=True)

import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from openmmtools_amd import testsystems as ts
from openmmtools_amd.system import system_to_desc
from openmmtools_amd._engine import HipEngine
KB = 0.008314462618153242
name = sys.argv[1] if len(sys.argv) > 1 else 'dhfr'
R = int(sys.argv[2]) if len(sys.argv) > 2 else 16
sysm = ts.DHFRExplicit() if name == 'dhfr' else ts.HostGuestExplicit()
box = np.diag(sysm.system.getDefaultPeriodicBoxVectors())
eng = HipEngine()
eng.set_system(system_to_desc(sysm.system, ewald_split=os.environ.get('DHFR_SPLIT', 'auto'))); eng.set_states(np.full(R, 1 / (KB * 300.0)))
eng.set_integrator('V R R O R R V', 0.002, 1.0, int(os.environ.get('DHFR_STEPS', '40')), True, 1e-8)
eng.set_replicas(R, 0, np.tile(sysm.positions, (R, 1, 1)), None, np.tile(box, (R, 1)), np.arange(R))
import time
for it in range(int(os.environ.get('DHFR_ITERS', '2'))):
    t0 = time.time(); eng.propagate(it); print('propagate', it, round((time.time() - t0) * 1e3, 1), 'ms')

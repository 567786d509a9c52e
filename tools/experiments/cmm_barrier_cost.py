import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from openmmtools_amd import testsystems as ts
from openmmtools_amd.system import system_to_desc
from openmmtools_amd._engine import HipEngine
KB = 0.008314462618153242
al = ts.AlanineDipeptideExplicit()
box = np.diag(al.system.getDefaultPeriodicBoxVectors())
R = 24
for cmm in (1, 0):
    eng = HipEngine()
    d = system_to_desc(al.system); d['cmm_frequency'] = cmm
    eng.set_system(d); eng.set_states(np.full(R, 1 / (KB * 300.0)))
    eng.set_integrator('V R R O R R V', 0.002, 1.0, 500, True, 1e-8)
    eng.set_replicas(R, 0, np.tile(al.positions, (R, 1, 1)), None, np.tile(box, (R, 1)), np.arange(R))
    eng.propagate(0)
    for it in range(1, 4):
        eng.propagate(it)
        print('cmm', cmm, 'gpu ms', eng.last_timing()['propagate_ms'])

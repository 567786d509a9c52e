"""Longer runs than the parity tests afford: temperature and constraint sanity of the headline system under the plain and a
multiple-time-step splitting, and of a multi-System ladder.  usage (GPU box): python tools/soak_check.py"""
import copy, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from openmmtools_amd import testsystems as ts
from openmmtools_amd.system import system_to_desc, NonbondedForce, HarmonicBondForce, HarmonicAngleForce, PeriodicTorsionForce
from openmmtools_amd._engine import HipEngine
KB = 0.008314462618153242
al = ts.AlanineDipeptideExplicit()
box = np.diag(al.system.getDefaultPeriodicBoxVectors())
ndof = 3 * 2269 - 2259 - 3
for name, splitting, groups in (('g-BAOAB', 'V R R O R R V', None), ('MTS bonded fast', 'V0 V1 R R O R R V1 R R O R R V1 V0', dict(bonded=1)),
                                ('MTS mesh slow', 'V2 V0 R R O R R V0 R R O R R V0 V2', dict(bonded=0, reciprocal=2))):
    system = copy.deepcopy(al.system)
    if groups:
        for f in system.getForces():
            if isinstance(f, (HarmonicBondForce, HarmonicAngleForce, PeriodicTorsionForce)):
                f.setForceGroup(groups.get('bonded', 0))
            elif isinstance(f, NonbondedForce):
                f.setForceGroup(0)
                if 'reciprocal' in groups:
                    f.setReciprocalSpaceForceGroup(groups['reciprocal'])
    R = 4
    eng = HipEngine()
    eng.set_system(system_to_desc(system)); eng.set_states(np.full(R, 1 / (KB * 300.0)))
    dt = 0.002 if 'MTS' not in name else 0.004
    eng.set_integrator(splitting, dt, 1.0, 500, True, 1e-8)
    eng.seed(7)
    eng.set_replicas(R, 0, np.tile(al.positions, (R, 1, 1)), None, np.tile(box, (R, 1)), np.arange(R))
    T = []
    for it in range(6):
        flags = eng.propagate(it)
        assert not flags.any(), (name, it, flags)
        ke = eng.get_replicas(positions=False, velocities=False, kinetic=True)[3]
        T.append(2.0 * ke / (ndof * KB))
    x = eng.get_replicas()[0]
    d = np.linalg.norm(x[:, 23, :] - x[:, 22, :], axis=1)          # an O-H bond of the first water (rigid)
    print('%-16s dt %.3f  T after each 500 steps (mean over %d replicas): %s   O-H %s' % (name, dt, R, np.round(np.mean(T, axis=1), 1), np.round(d, 5)))
    eng.close()

"""Converts the reference's Amber input files into compact .npz system descriptions.

Run in the build container only (needs /root/reference):  python tools/convert_amber.py
Outputs openmmtools_amd/data/{alanine-dipeptide-explicit,cb7-b2-explicit,dhfr-explicit,alanine-dipeptide-vacuum,cb7-b2-vacuum}.npz, the inputs of
testsystems.AlanineDipeptideExplicit / HostGuestExplicit / DHFRExplicit (reference:
openmmtools/testsystems.py:3499-3527, 3821-3857, 3895-3923).  The reference's data files are the
physical input of the benchmark configs; only derived numeric arrays are stored.
"""
import os
import sys
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from openmmtools_amd import amber   # noqa: E402
from openmmtools_amd.system import HarmonicBondForce, HarmonicAngleForce, PeriodicTorsionForce   # noqa: E402

REF = '/root/reference/openmmtools/data'
JOBS = [
    ('alanine-dipeptide-explicit', 'alanine-dipeptide-explicit/alanine-dipeptide.prmtop',
     'alanine-dipeptide-explicit/alanine-dipeptide.crd'),
    ('cb7-b2-explicit', 'cb7-b2/complex-explicit.prmtop', 'cb7-b2/complex-explicit.inpcrd'),
    ('dhfr-explicit', 'dhfr/JAC.prmtop', 'dhfr/JAC.inpcrd'),
    # testsystems.AlanineDipeptideVacuum (testsystems.py:3352-3388): the same parameter set without solvent, NoCutoff, no box
    ('alanine-dipeptide-vacuum', 'alanine-dipeptide-gbsa/alanine-dipeptide.prmtop', 'alanine-dipeptide-gbsa/alanine-dipeptide.crd'),
    # testsystems.HostGuestVacuum (testsystems.py:3660-3712): CB7:B2 without solvent
    ('cb7-b2-vacuum', 'cb7-b2/complex-vacuum.prmtop', 'cb7-b2/complex-vacuum.inpcrd'),
]

for name, top, crd in JOBS:
    prm = amber.read_prmtop(os.path.join(REF, top))
    system, nb = amber.create_system(prm)
    n = system.getNumParticles()
    pos, vel, box = amber.read_inpcrd(os.path.join(REF, crd), n)
    if box is None:
        box = np.zeros(3)
    forces = {type(f).__name__: f for f in system.getForces()}
    bf, af, tf = forces['HarmonicBondForce'], forces['HarmonicAngleForce'], forces['PeriodicTorsionForce']
    p = np.array(nb.particles)
    out = dict(
        mass=np.array(system.masses), charge=p[:, 0], sigma=p[:, 1], epsilon=p[:, 2], box=box,
        bond_atoms=np.array([b[:2] for b in bf.bonds], dtype=np.int32).reshape(-1, 2),
        bond_params=np.array([b[2:] for b in bf.bonds]).reshape(-1, 2),
        angle_atoms=np.array([a[:3] for a in af.angles], dtype=np.int32).reshape(-1, 3),
        angle_params=np.array([a[3:] for a in af.angles]).reshape(-1, 2),
        torsion_atoms=np.array([t[:4] for t in tf.torsions], dtype=np.int32).reshape(-1, 4),
        torsion_params=np.array([t[4:] for t in tf.torsions], dtype=np.float64).reshape(-1, 3),
        exception_atoms=np.array([e[:2] for e in nb.exceptions], dtype=np.int32).reshape(-1, 2),
        exception_params=np.array([e[2:] for e in nb.exceptions]).reshape(-1, 3),
        constraint_atoms=np.array([c[:2] for c in system.constraints], dtype=np.int32).reshape(-1, 2),
        constraint_dist=np.array([c[2] for c in system.constraints]),
        positions=pos.astype(np.float64),
        residue_names=np.array(prm['RESIDUE_LABEL']),
        residue_pointer=np.array(prm['RESIDUE_POINTER'], dtype=np.int32),
    )
    if vel is not None:
        out['velocities'] = vel
    if 'RADII' in prm and 'SCREEN' in prm:
        # Generalized-Born radii (nm) and screening factors of the topology (prmtop RADIUS_SET): what prmtop.createSystem(implicitSolvent=...) hands
        # GBSAOBCForce.addParticle(charge, radius, scale)
        out['gb_radii'] = np.array(prm['RADII'][:n], dtype=np.float64) * 0.1
        out['gb_screen'] = np.array(prm['SCREEN'][:n], dtype=np.float64)
    path = os.path.join(ROOT, 'openmmtools_amd', 'data', name + '.npz')
    np.savez_compressed(path, **out)
    print(name, 'atoms', n, 'bonds', len(bf.bonds), 'angles', len(af.angles), 'torsions', len(tf.torsions),
          'exceptions', len(nb.exceptions), 'constraints', len(system.constraints), 'box', box,
          'net charge %.4f' % p[:, 0].sum(), 'size %.0f KB' % (os.path.getsize(path) / 1024))

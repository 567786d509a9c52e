"""The benchmark test systems of BASELINE.json, built on the System shim.

Mirrors the constructor logic of openmmtools/testsystems.py for
  HarmonicOscillator        :685-802
  LennardJonesFluid         :1872-2030 (+ subrandom_particle_positions :236-289 and the Sobol'
                            generator it calls, sobol.i4_sobol_generate(3, N, 1))
  AlanineDipeptideExplicit  :3465-3527   HostGuestExplicit :3789-3857   DHFRExplicit :3863-3923
The Amber-built systems are loaded from compact .npz system descriptions under
openmmtools_amd/data/, produced from the reference's prmtop/inpcrd files by
tools/convert_amber.py with the parser in openmmtools_amd/amber.py.
"""
import os
import numpy as np
from . import unit
from .system import (System, NonbondedForce, CustomExternalForce, HarmonicBondForce, HarmonicAngleForce,
                     PeriodicTorsionForce, CMMotionRemover)

DEFAULT_EWALD_ERROR_TOLERANCE = 1.0e-5            # testsystems.py:69
DEFAULT_CUTOFF_DISTANCE = 10.0 * unit.angstroms   # :70
DEFAULT_SWITCH_WIDTH = 1.5 * unit.angstroms       # :71

_DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'data')


class TestSystem:
    __test__ = False     # not a pytest class

    def __init__(self, **kwargs):
        self.system = None
        self.positions = None
        self.topology = None


class HarmonicOscillator(TestSystem):
    """3D harmonic oscillator: one particle in U = K/2 ((x-x0)^2 + y^2 + z^2) + U0 (testsystems.py:761-802)."""

    def __init__(self, K=100.0 * unit.kilocalories_per_mole / unit.angstroms ** 2, mass=39.948 * unit.amu,
                 U0=0.0 * unit.kilojoules_per_mole, **kwargs):
        super().__init__(**kwargs)
        system = System()
        system.addParticle(mass)
        positions = np.zeros([1, 3], np.float32).astype(np.float64)
        edge = 1000.0 * unit.nanometers
        system.setDefaultPeriodicBoxVectors([edge, 0, 0], [0, edge, 0], [0, 0, edge])
        force = CustomExternalForce(CustomExternalForce.HARMONIC_EXPRESSION)
        force.addGlobalParameter('testsystems_HarmonicOscillator_K', K)
        force.addGlobalParameter('testsystems_HarmonicOscillator_x0', 0.0)
        force.addGlobalParameter('testsystems_HarmonicOscillator_U0', U0)
        force.addParticle(0, [])
        system.addForce(force)
        self.K, self.mass, self.U0 = K, mass, U0
        self.system, self.positions = system, positions
        self.ndof = 3

    def get_potential_expectation(self, state):
        from .constants import kB
        return 1.5 * kB * state.temperature          # testsystems.py:820


def sobol3(n_points):
    """First n points (seeds 0..n-1) of the 3-D Sobol' sequence in Gray-code order, 30-bit.

    Equivalent to the reference's sobol.i4_sobol_generate(3, n, 1) (sobol.py:136-169, 171-436):
    Bratley-Fox direction numbers, dimension 1 = van der Corput, dimension 2 from x+1 with m = (1),
    dimension 3 from x^2+x+1 with m = (1, 1).  Checked against tests/golden/sobol_512x3.npy.
    """
    bits = 30
    m = np.zeros((3, bits), dtype=np.int64)
    m[0, :] = 1
    m[1, 0] = 1
    for i in range(1, bits):
        m[1, i] = (2 * m[1, i - 1]) ^ m[1, i - 1]
    m[2, 0] = m[2, 1] = 1
    for i in range(2, bits):
        m[2, i] = (2 * m[2, i - 1]) ^ (4 * m[2, i - 2]) ^ m[2, i - 2]
    v = m * (2 ** (bits - 1 - np.arange(bits)))[None, :]          # direction numbers scaled to 2^30
    out = np.zeros((3, n_points))
    q = np.zeros(3, dtype=np.int64)
    for n in range(n_points):
        out[:, n] = q / float(2 ** bits)
        l = 0                                # position of the lowest zero bit of n
        k = n
        while k & 1:
            k >>= 1
            l += 1
        q ^= v[:, l]
    return out


def subrandom_particle_positions(nparticles, box_vectors):
    """testsystems.py:236-289 (method='sobol'): float32 positions, x[dim] * L[dim]."""
    x = np.array(sobol3(nparticles), np.float32)
    positions = np.zeros([nparticles, 3], np.float32)
    for dim in range(3):
        l = np.float32(box_vectors[dim][dim])
        positions[:, dim] = x[dim, :] * l
    return positions.astype(np.float64)


class LennardJonesFluid(TestSystem):
    """Periodic argon-like LJ fluid (testsystems.py:1939-2030): CutoffPeriodic (zero charge), switching
    function from cutoff - switch_width, analytic long-range dispersion correction, Sobol' positions."""

    def __init__(self, nparticles=1000, reduced_density=0.05, mass=39.9 * unit.amu, sigma=3.4 * unit.angstrom,
                 epsilon=0.238 * unit.kilocalories_per_mole, cutoff=None, switch_width=3.4 * unit.angstrom,
                 shift=False, dispersion_correction=True, lattice=False, charge=None, ewaldErrorTolerance=None,
                 **kwargs):
        super().__init__(**kwargs)
        if shift or lattice or charge is not None:
            raise NotImplementedError('shift / lattice / charged LJ fluids are not part of the benchmark configs')
        if cutoff is None:
            cutoff = 3.0 * sigma                                           # :1957-1958
        system = System()
        number_density = reduced_density / sigma ** 3                      # :1970
        volume = nparticles / number_density
        box_edge = volume ** (1.0 / 3.0)
        system.setDefaultPeriodicBoxVectors([box_edge, 0, 0], [0, box_edge, 0], [0, 0, box_edge])
        nb = NonbondedForce()
        nb.setNonbondedMethod(NonbondedForce.CutoffPeriodic)
        nb.setCutoffDistance(cutoff)
        nb.setUseDispersionCorrection(dispersion_correction)
        nb.setUseSwitchingFunction(False)
        if switch_width is not None:
            nb.setUseSwitchingFunction(True)
            nb.setSwitchingDistance(cutoff - switch_width)                 # :1987-1989
        for _ in range(nparticles):
            system.addParticle(mass)
            nb.addParticle(0.0, sigma, epsilon)
        positions = subrandom_particle_positions(nparticles, system.getDefaultPeriodicBoxVectors())
        system.addForce(nb)
        self.system, self.positions = system, positions
        self.ndof = 3 * nparticles


class IdealGas(TestSystem):
    """testsystems.py:2631-2735: non-interacting particles in a periodic box of volume N kT / p with a null periodic
    NonbondedForce (so that a barostat can act): <V> = (N + 1) kT / p under the Monte Carlo barostat, <U> = 0."""

    def __init__(self, nparticles=216, mass=39.9 * unit.amu, temperature=298.0 * unit.kelvin, pressure=1.0 * unit.atmosphere,
                 volume=None, **kwargs):
        super().__init__(**kwargs)
        from . import constants
        if volume is None:
            volume = nparticles * constants.kB * float(unit.to_md(temperature)) / float(unit.to_md(pressure))      # nm^3, :2664-2665
        length = float(unit.to_md(volume)) ** (1.0 / 3.0)
        system = System()
        system.setDefaultPeriodicBoxVectors([length, 0, 0], [0, length, 0], [0, 0, length])
        nb = NonbondedForce()
        nb.setNonbondedMethod(NonbondedForce.CutoffPeriodic)                # :2678-2682: charge 0, sigma 1 nm, epsilon 0
        nb.setCutoffDistance(min(1.0, 0.45 * length))
        nb.setUseDispersionCorrection(False)
        for _ in range(nparticles):
            system.addParticle(mass)
            nb.addParticle(0.0, 1.0, 0.0)
        system.addForce(nb)
        self.system, self.positions = system, subrandom_particle_positions(nparticles, system.getDefaultPeriodicBoxVectors())
        self.ndof = 3 * nparticles


def _load_npz_system(name):
    path = os.path.join(_DATA, name + '.npz')
    if not os.path.exists(path):
        raise FileNotFoundError('%s missing: run tools/convert_amber.py where the reference data exists' % path)
    z = np.load(path)
    system = System()
    for m in z['mass']:
        system.addParticle(float(m))
    box = z['box']
    system.setDefaultPeriodicBoxVectors([box[0], 0, 0], [0, box[1], 0], [0, 0, box[2]])
    bf = HarmonicBondForce()
    for (i, j), (r0, k) in zip(z['bond_atoms'], z['bond_params']):
        bf.addBond(i, j, r0, k)
    af = HarmonicAngleForce()
    for (i, j, k3), (th, k) in zip(z['angle_atoms'], z['angle_params']):
        af.addAngle(i, j, k3, th, k)
    tf = PeriodicTorsionForce()
    for (i, j, k3, l), (n, ph, k) in zip(z['torsion_atoms'], z['torsion_params']):
        tf.addTorsion(i, j, k3, l, int(n), ph, k)
    nb = NonbondedForce()
    for q, s, e in zip(z['charge'], z['sigma'], z['epsilon']):
        nb.addParticle(q, s, e)
    nb.exceptions = [(int(a), int(b), float(p[0]), float(p[1]), float(p[2]))
                     for (a, b), p in zip(z['exception_atoms'], z['exception_params'])]
    for (i, j), d in zip(z['constraint_atoms'], z['constraint_dist']):
        system.addConstraint(i, j, d)
    for f in (bf, af, tf, nb):
        system.addForce(f)
    system.addForce(CMMotionRemover(1))            # prmtop.createSystem default removeCMMotion=True
    positions = z['positions'].astype(np.float64)
    velocities = z['velocities'].astype(np.float64) if 'velocities' in z.files else None
    return system, nb, positions, velocities, z


class _AmberExplicit(TestSystem):
    _name = None

    def __init__(self, constraints='HBonds', rigid_water=True, nonbondedCutoff=DEFAULT_CUTOFF_DISTANCE,
                 use_dispersion_correction=True, nonbondedMethod='PME', hydrogenMass=None,
                 switch_width=DEFAULT_SWITCH_WIDTH, ewaldErrorTolerance=DEFAULT_EWALD_ERROR_TOLERANCE, **kwargs):
        super().__init__(**kwargs)
        if constraints != 'HBonds' or not rigid_water or hydrogenMass is not None:
            raise NotImplementedError('only constraints=HBonds, rigid_water=True, no HMR (the testsystem defaults)')
        system, nb, positions, velocities, z = _load_npz_system(self._name)
        method = {'PME': NonbondedForce.PME, 'CutoffPeriodic': NonbondedForce.CutoffPeriodic}[nonbondedMethod]
        nb.setNonbondedMethod(method)
        nb.setCutoffDistance(nonbondedCutoff)
        nb.setUseDispersionCorrection(use_dispersion_correction)
        nb.setEwaldErrorTolerance(ewaldErrorTolerance)
        if switch_width is not None:                                       # testsystems.py:3515-3517
            nb.setUseSwitchingFunction(True)
            nb.setSwitchingDistance(nonbondedCutoff - switch_width)
        self.system, self.positions, self.velocities = system, positions, velocities
        self.residue_names = [str(s) for s in z['residue_names']] if 'residue_names' in z.files else None


class AlanineDipeptideExplicit(_AmberExplicit):
    """testsystems.py:3465-3527: ACE-ALA-NME + 749 TIP3P waters, 2269 atoms, PME."""
    _name = 'alanine-dipeptide-explicit'


class AlanineDipeptideVacuum(TestSystem):
    """testsystems.py:3352-3388: ACE-ALA-NME (22 atoms) without solvent -- prmtop.createSystem(implicitSolvent=None, constraints=HBonds,
    nonbondedCutoff=None): NonbondedForce with NoCutoff, no periodic box, CMMotionRemover."""

    def __init__(self, constraints='HBonds', hydrogenMass=None, **kwargs):
        super().__init__(**kwargs)
        if constraints != 'HBonds' or hydrogenMass is not None:
            raise NotImplementedError('only constraints=HBonds, no HMR (the testsystem defaults)')
        system, nb, positions, velocities, z = _load_npz_system('alanine-dipeptide-vacuum')
        nb.setNonbondedMethod(NonbondedForce.NoCutoff)
        self.system, self.positions, self.velocities = system, positions, None
        self.residue_names = [str(s) for s in z['residue_names']] if 'residue_names' in z.files else None


class HostGuestVacuum(TestSystem):
    """testsystems.py:3660-3712: CB7 host + B2 guest (156 atoms) without solvent -- prmtop.createSystem(implicitSolvent=None,
    constraints=HBonds, nonbondedMethod=NoCutoff)."""

    def __init__(self, **kwargs):
        super().__init__(**kwargs)
        system, nb, positions, velocities, z = _load_npz_system('cb7-b2-vacuum')
        nb.setNonbondedMethod(NonbondedForce.NoCutoff)
        self.system, self.positions, self.velocities = system, positions, None
        self.residue_names = [str(s) for s in z['residue_names']] if 'residue_names' in z.files else None


class AlanineDipeptideImplicit(AlanineDipeptideVacuum):
    """testsystems.py:3424-3462: the vacuum dipeptide + Generalized-Born implicit solvent.  The reference's default is app.OBC1, which
    current OpenMM builds as a CustomGBForce from its own expression library (not in the reference's tree); what is built here is
    ``implicitSolvent='OBC2'`` = openmm.GBSAOBCForce -- the force the reference's alchemical factory spells out term by term
    (alchemy.py:2144-2225) -- with the topology's own GB radii and screening factors (prmtop RADII / SCREEN, mbondi2), solvent
    dielectric 78.5, solute dielectric 1, ACE surface term."""

    def __init__(self, implicitSolvent='OBC2', constraints='HBonds', hydrogenMass=None, **kwargs):
        if implicitSolvent != 'OBC2':
            raise NotImplementedError("implicitSolvent=%r: only 'OBC2' (openmm.GBSAOBCForce) is built" % (implicitSolvent,))
        super().__init__(constraints=constraints, hydrogenMass=hydrogenMass, **kwargs)
        from .system import GBSAOBCForce
        z = np.load(os.path.join(_DATA, 'alanine-dipeptide-vacuum.npz'))
        gb = GBSAOBCForce()
        for q, r, sc in zip(z['charge'], z['gb_radii'], z['gb_screen']):
            gb.addParticle(q, r, sc)
        self.system.addForce(gb)


class HostGuestExplicit(_AmberExplicit):
    """testsystems.py:3789-3857: CB7 + B2 guest + 1445 TIP3P waters, 4491 atoms."""
    _name = 'cb7-b2-explicit'


class DHFRExplicit(_AmberExplicit):
    """testsystems.py:3863-3923: DHFR (JAC benchmark), 23558 atoms, positions and velocities from JAC.inpcrd."""
    _name = 'dhfr-explicit'

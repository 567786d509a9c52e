"""libremd_cpu.so (oracle/cpu/remd_cpu.cpp): the CPU implementation of include/remd_hip.h that bench.py times as the CPU
baseline (SURVEY 8(b) last line, 8(d)).  It is driven here through the SAME ctypes class as the
HIP library (openmmtools_amd._engine.HipEngine with lib_path) and checked against the independent f64 Python oracle
(autograd forces, scipy neighbour search, numpy FFT): energies to 1e-9 relative, forces to 1e-7 of the largest force,
short trajectories to 1e-8 nm, mixing bit-exact.  CPU-only tests: this is also the check that the C ABI as declared in
include/remd_hip.h can be implemented and driven end to end without a GPU."""
import os
import numpy as np
import pytest
import oracle
from openmmtools_amd import testsystems as ts, alchemy, states, mcmc, unit
from openmmtools_amd.system import system_to_desc
from openmmtools_amd._engine import HipEngine
from oracle import md_oracle as mo
from oracle.forcefield import ForceFieldOracle
from oracle_engine import OracleEngine

KB = 0.008314462618153242
SEED = 0xC0FFEE
CPU_LIB = os.path.join(os.path.dirname(os.path.abspath(oracle.__file__)), '_build', 'libremd_cpu.so')


@pytest.fixture(scope='module')
def cpu_engine_factory():
    if not os.path.exists(CPU_LIB):
        oracle.build()
    made = []

    def make():
        e = HipEngine(lib_path=CPU_LIB)
        made.append(e)
        return e
    yield make
    for e in made:
        e.close()


def _setup(eng, system, positions, R=2, temperature=300.0, lam_s=None, lam_e=None, jitter=0.0, splitting='V R O R V', dt=0.001,
           n_steps=5, labels=None, econst=None, reassign=True):
    desc = system_to_desc(system)
    eng.set_system(desc)
    K = R if lam_s is None else len(lam_s)
    eng.set_states(np.full(K, 1.0 / (KB * temperature)), lam_s, lam_e, econst)
    eng.set_integrator(splitting, dt, 1.0, n_steps, reassign, 1e-8)
    eng.seed(SEED)
    rng = np.random.default_rng(3)
    x = np.stack([positions + jitter * rng.normal(size=positions.shape) * (r > 0) for r in range(R)])
    box = np.tile(np.diag(system.getDefaultPeriodicBoxVectors()), (R, 1))
    eng.set_replicas(R, 0, x, None, box, np.arange(R) % K if labels is None else labels)
    return desc, x, box


@pytest.mark.parametrize('shape', [(8, 8, 8), (16, 24, 40), (75, 75, 72), (45, 50, 27)])
def test_cpu_fft3d_matches_numpy(cpu_engine_factory, shape):
    eng = cpu_engine_factory()
    rng = np.random.default_rng(sum(shape))
    a = (rng.normal(size=shape) + 1j * rng.normal(size=shape)).astype(np.complex64)
    ref = np.fft.fftn(a.astype(np.complex128))
    got = eng.test_fft3d(a)
    assert np.abs(got - ref).max() < 3e-7 * np.abs(ref).max()          # f64 transform, f32 at the boundary
    back = eng.test_fft3d(got, inverse=True) / np.prod(shape)
    assert np.abs(back - a).max() < 1e-6 * np.abs(a).max()


def test_cpu_lj_fluid_energy_forces_and_alchemical_rows(cpu_engine_factory):
    lj = ts.LennardJonesFluid(nparticles=512)
    region = alchemy.AlchemicalRegion(alchemical_atoms=range(10))
    system = alchemy.AbsoluteAlchemicalFactory().create_alchemical_system(lj.system, region)
    lam = np.linspace(1.0, 0.0, 16)
    nb = [f for f in system.getForces() if hasattr(f, 'particles') and hasattr(f, 'exceptions')][0]
    V = np.prod(np.diag(system.getDefaultPeriodicBoxVectors()))
    econst = alchemy.alchemical_long_range_constants(system, nb, lam, V)
    eng = cpu_engine_factory()
    desc, x, box = _setup(eng, system, lj.positions, R=3, lam_s=lam, jitter=0.01, labels=[0, 7, 15], econst=econst)
    ff = ForceFieldOracle(desc)
    rows, U = eng.compute_energies(want_potential=True)
    f = eng.get_forces()
    beta = 1.0 / (KB * 300.0)
    for r, k in enumerate([0, 7, 15]):
        ref = beta * (ff.state_energies(x[r], box[r], lam, np.ones(16)) + econst)
        assert np.allclose(rows[r], ref, rtol=1e-9, atol=1e-9), np.abs(rows[r] - ref).max()
        e_ref, f_ref = ff.energy_forces(x[r], box[r], lambda_sterics=lam[k])
        assert np.isclose(U[r], e_ref, rtol=1e-10, atol=1e-9)
        assert np.abs(f[r] - f_ref).max() < 1e-8 * max(1.0, np.abs(f_ref).max())


@pytest.fixture(scope='module')
def alanine():
    al = ts.AlanineDipeptideExplicit()
    return al, system_to_desc(al.system)


def test_cpu_alanine_components_and_forces(cpu_engine_factory, alanine):
    """Bonded terms, LJ + Ewald direct space on exact Verlet lists, 1-4 exceptions, exclusion correction, smooth PME
    (in-tree mixed-radix FFT, half spectrum), self / background / dispersion constants."""
    al, _ = alanine
    eng = cpu_engine_factory()
    desc, x, box = _setup(eng, al.system, al.positions, R=2, jitter=0.002)
    ff = ForceFieldOracle(desc)
    U = eng.compute_energies(want_potential=True)[1]
    comp = np.zeros((2, 9))
    eng._check(eng.lib.remd_get_energy_components(eng.h, comp.ctypes.data_as(__import__('ctypes').POINTER(__import__('ctypes').c_double))), 'components')
    f = eng.get_forces()
    for r in range(2):
        e_ref, f_ref = ff.energy_forces(x[r], box[r])
        assert np.isclose(U[r], e_ref, rtol=1e-9), (U[r], e_ref)
        assert np.isclose(comp[r].sum(), U[r], rtol=1e-12)
        assert np.abs(f[r] - f_ref).max() < 1e-7 * np.abs(f_ref).max(), np.abs(f[r] - f_ref).max()
    import torch
    xt = torch.tensor(x[0])
    assert np.isclose(comp[0, 1] + comp[0, 2] + comp[0, 3], float(ff._bonded(xt)), rtol=1e-10)
    q = torch.tensor(np.asarray(desc['charge'], dtype=np.float64))
    assert np.isclose(comp[0, 6], float(ff.pme_reciprocal(xt, torch.tensor(box[0]), q)), rtol=1e-9)


def test_cpu_alanine_propagation_tracks_oracle(cpu_engine_factory, alanine):
    """g-BAOAB with SHAKE/RATTLE clusters, CM-motion removal and Maxwell-Boltzmann reassignment: 6 steps of 2 fs."""
    al, _ = alanine
    eng = cpu_engine_factory()
    desc, x, box = _setup(eng, al.system, al.positions, R=2, jitter=0.002, splitting='V R R O R R V', dt=0.002, n_steps=6)
    ora = OracleEngine(ForceFieldOracle)
    ora.set_system(desc)
    ora.set_states(np.full(2, 1.0 / (KB * 300.0)))
    ora.set_integrator('V R R O R R V', 0.002, 1.0, 6, True)
    ora.seed(SEED)
    ora.set_replicas(2, 0, x, None, box, np.arange(2))
    assert not eng.propagate(3).any() and not ora.propagate(3).any()
    xg, vg, _, ke = eng.get_replicas(kinetic=True)
    assert np.abs(xg - ora.x).max() < 1e-8 and np.abs(vg - ora.v).max() < 1e-6
    assert np.allclose(ke, [mo.kinetic_energy(ora.sys.mass, ora.v[r]) for r in range(2)], rtol=1e-7)
    cons = mo.OracleSystem(desc).constraints
    i, j, dist = np.array([c[0] for c in cons]), np.array([c[1] for c in cons]), np.array([c[2] for c in cons])
    assert np.abs(np.linalg.norm(xg[0][j] - xg[0][i], axis=1) - dist).max() < 1e-10


def test_cpu_hostguest_lambda_rows(cpu_engine_factory):
    """lambda_electrostatics (exact PME, quadratic in lambda) and lambda_sterics (soft core) states of CB7:B2."""
    hg = ts.HostGuestExplicit()
    region = alchemy.AlchemicalRegion(alchemical_atoms=range(126, 156))
    system = alchemy.AbsoluteAlchemicalFactory().create_alchemical_system(hg.system, region)
    lam_e = np.array([1.0, 0.3, 0.0, 0.0, 0.0])
    lam_s = np.array([1.0, 1.0, 1.0, 0.4, 0.0])
    eng = cpu_engine_factory()
    desc, x, box = _setup(eng, system, hg.positions, R=1, lam_s=lam_s, lam_e=lam_e, labels=[1])
    rows = eng.compute_energies()
    ff = ForceFieldOracle(desc)
    ref = ff.state_energies(x[0], box[0], lam_s, lam_e) / (KB * 300.0)
    assert np.allclose(rows[0], ref, rtol=1e-9), np.abs(rows[0] / ref - 1).max()
    f = eng.get_forces()
    f_ref = ff.energy_forces(x[0], box[0], lambda_sterics=1.0, lambda_electrostatics=0.3)[1]
    assert np.abs(f[0] - f_ref).max() < 1e-7 * np.abs(f_ref).max()


def test_cpu_dhfr_energy_and_forces_two_independent_implementations(cpu_engine_factory):
    """Config 5's system (DHFR, 23 558 atoms, PME at the reference split) has no reference-held energy: its parameters are pinned to
    the reference's Amber files (tests/test_testsystem_defaults.py), and its energy and forces are computed here by two implementations
    that share no code -- the torch autograd oracle (pair search + smooth PME with torch.fft) and the C++ port (cell lists, hand-derived
    forces, in-tree FFT): 1e-11 relative on the energy, 1e-10 of the largest force.  The device is held to the oracle by the -m gpu tests."""
    dh = ts.DHFRExplicit()
    eng = cpu_engine_factory()
    desc, x, box = _setup(eng, dh.system, dh.positions, R=1)
    u = float(np.asarray(eng.compute_energies()).ravel()[0]) * (KB * 300.0)
    e, f = ForceFieldOracle(desc).energy_forces(x[0], box[0])
    assert abs(u - e) < 1e-11 * abs(e), (u, e)
    assert np.abs(eng.get_forces()[0] - f).max() < 1e-10 * np.abs(f).max()


def test_cpu_annihilated_sterics(cpu_engine_factory):
    """AlchemicalRegion(annihilate_sterics=True) (alchemy.py:421, 1767-1779, 1841-1846; remd_set_alchemical_options): the
    Lennard-Jones pairs and 1-4 exceptions INSIDE the alchemical region are soft-core and lambda-controlled too.  CB7:B2 guest:
    u_kl rows, own-state potential and forces of the C++ port against the torch oracle (pinned to the reference's expressions by
    the document interpreter of tests/test_alchemical_store_cpu.py)."""
    hg = ts.HostGuestExplicit()
    system = alchemy.AbsoluteAlchemicalFactory().create_alchemical_system(
        hg.system, alchemy.AlchemicalRegion(alchemical_atoms=range(126, 156), annihilate_sterics=True))
    lam_e = np.array([1.0, 0.3, 0.0, 0.0, 0.0])
    lam_s = np.array([1.0, 1.0, 1.0, 0.4, 0.0])
    eng = cpu_engine_factory()
    desc, x, box = _setup(eng, system, hg.positions, R=1, lam_s=lam_s, lam_e=lam_e, labels=[3])
    assert desc['annihilate_sterics'] is True
    ff = ForceFieldOracle(desc)
    rows, U = eng.compute_energies(want_potential=True)
    ref = ff.state_energies(x[0], box[0], lam_s, lam_e)
    assert np.allclose(rows[0], ref / (KB * 300.0), rtol=1e-9), np.abs(rows[0] * KB * 300.0 / ref - 1).max()
    assert np.isclose(U[0], ref[3], rtol=1e-9)
    decoupled = ForceFieldOracle(system_to_desc(alchemy.AbsoluteAlchemicalFactory().create_alchemical_system(
        hg.system, alchemy.AlchemicalRegion(alchemical_atoms=range(126, 156)))))
    assert abs(ref[4] - decoupled.state_energies(x[0], box[0], lam_s, lam_e)[4]) > 10.0
    f = eng.get_forces()
    f_ref = ff.energy_forces(x[0], box[0], lambda_sterics=0.4, lambda_electrostatics=0.0)[1]
    assert np.abs(f[0] - f_ref).max() < 1e-7 * np.abs(f_ref).max()


def test_cpu_softcore_exceptions_of_a_region_that_cuts_a_molecule(cpu_engine_factory, alanine):
    """Round 4: 1-4 exceptions between an alchemical and a non-alchemical atom carry soft-core, lambda_sterics-controlled
    Lennard-Jones (the factory's CustomBondForce, alchemy.py:1836-1851, 1985-1998); earlier rounds left them at full strength.
    Alanine dipeptide with only its first 10 atoms alchemical: 16 such exceptions.  u_kl rows, the own-state potential and the
    forces of the C++ port against the torch oracle (which tests/test_alchemical_store_cpu.py pins against the reference's
    energy expression)."""
    alanine = alanine[0]
    system = alchemy.AbsoluteAlchemicalFactory().create_alchemical_system(alanine.system, alchemy.AlchemicalRegion(alchemical_atoms=range(10)))
    lam_e = np.array([1.0, 0.5, 0.0, 0.0, 0.0])
    lam_s = np.array([1.0, 1.0, 1.0, 0.45, 0.0])
    eng = cpu_engine_factory()
    desc, x, box = _setup(eng, system, alanine.positions, R=1, lam_s=lam_s, lam_e=lam_e, labels=[3])
    ff = ForceFieldOracle(desc)
    nb_exc = [(i, j) for (i, j), p in zip(desc['exception_atoms'], desc['exception_params']) if p[2] != 0 and (i < 10) != (j < 10)]
    assert len(nb_exc) == 16
    rows, U = eng.compute_energies(want_potential=True)
    ref = ff.state_energies(x[0], box[0], lam_s, lam_e)
    assert np.allclose(rows[0], ref / (KB * 300.0), rtol=1e-9), np.abs(rows[0] * KB * 300.0 / ref - 1).max()
    assert np.isclose(U[0], ref[3], rtol=1e-9)
    assert abs(ref[3] - ff.energy_forces(x[0], box[0], lambda_sterics=1.0, lambda_electrostatics=0.0, forces=False)[0]) > 1.0   # lambda matters
    f = eng.get_forces()
    f_ref = ff.energy_forces(x[0], box[0], lambda_sterics=0.45, lambda_electrostatics=0.0)[1]
    assert np.abs(f[0] - f_ref).max() < 1e-7 * np.abs(f_ref).max()


def test_cpu_copy_replicas_between_handles(cpu_engine_factory):
    """remd_copy_replicas (one handle per compatibility group, multistate/_engine_pool.py): rows go from one handle's slots to
    another's without the host; a handle sized with x = NULL and filled that way evaluates like one given the coordinates."""
    lj = ts.LennardJonesFluid(nparticles=64)
    a, b, c = cpu_engine_factory(), cpu_engine_factory(), cpu_engine_factory()
    desc, x, box = _setup(a, lj.system, lj.positions, R=4, jitter=0.01)
    v = np.random.default_rng(1).normal(size=x.shape)
    box = box * np.array([1.0, 1.01, 1.02, 1.03])[:, None]
    a.set_replicas(4, 0, x, v, box, np.arange(4))
    for eng in (b, c):
        eng.set_system(desc)
        eng.set_states(np.full(4, 1.0 / (KB * 300.0)))
        eng.set_integrator('V R O R V', 0.001, 1.0, 5, True, 1e-8)
        eng.seed(SEED)
    b.set_replicas(2, 0, None, None, box[:2], np.arange(2))             # sized only
    b.copy_replicas([1, 0], a, [3, 1], 7)
    xb, vb, _, _ = b.get_replicas()
    assert np.array_equal(xb, x[[1, 3]]) and np.array_equal(vb, v[[1, 3]]) and np.array_equal(b.get_boxes(), box[[1, 3]])
    c.set_replicas(2, 0, x[[1, 3]], v[[1, 3]], box[[1, 3]], np.arange(2))
    assert np.array_equal(b.compute_energies(), c.compute_energies()) and np.array_equal(b.get_forces(), c.get_forces())
    b.copy_replicas([0], a, [0], 1)                                     # positions only
    xb2, vb2, _, _ = b.get_replicas()
    assert np.array_equal(xb2[0], x[0]) and np.array_equal(vb2, vb) and np.array_equal(b.get_boxes(), box[[1, 3]])
    for bad in (([0, 0], [1, 2]), ([2], [0]), ([0], [4])):
        with pytest.raises(RuntimeError, match='remd_copy_replicas'):
            b.copy_replicas(bad[0], a, bad[1], 7)
    with pytest.raises(RuntimeError, match='same handle'):
        a.copy_replicas([0], a, [1], 7)


def test_cpu_sampler_equals_python_oracle_engine(cpu_engine_factory):
    """The whole iteration (mix -> propagate -> u_kl) of ParallelTemperingSampler through the ABI on the CPU library equals
    the run on the Python OracleEngine: labels and count matrices exactly, energies to round-off."""
    from openmmtools_amd.multistate import ParallelTemperingSampler
    ho = ts.HarmonicOscillator()
    th = states.ThermodynamicState(ho.system, 300.0)
    ss = states.SamplerState(ho.positions)
    move = mcmc.LangevinSplittingDynamicsMove(timestep=1.0 * unit.femtosecond, collision_rate=1.0 / unit.picosecond,
                                              n_steps=20, reassign_velocities=True, splitting='V R O R V')
    out = []
    for engine in (cpu_engine_factory(), OracleEngine()):
        engine.is_device = False
        s = ParallelTemperingSampler(mcmc_moves=move, number_of_iterations=5, engine=engine, seed=7)
        s.create(th, [ss], min_temperature=300.0, max_temperature=600.0, n_temperatures=4)
        hist = []
        for _ in range(5):
            s.run(1)
            hist.append((s.replica_thermodynamic_states.copy(), s.energy_thermodynamic_states.copy(), s._n_proposed_matrix.copy()))
        out.append(hist)
    for (la, ua, pa), (lb, ub, pb) in zip(*out):
        assert np.array_equal(la, lb) and np.array_equal(pa, pb)
        assert np.allclose(ua, ub, rtol=1e-9, atol=1e-12)


def test_cpu_barostat_tracks_the_python_oracle(cpu_engine_factory):
    """NPT: Monte Carlo barostat inside the Langevin step (frequency 5 here) and explicit volume moves; boxes, volumes and
    the beta p V term of u_kl against OracleEngine on the same Philox stream."""
    lj = ts.LennardJonesFluid(nparticles=216)
    desc = system_to_desc(lj.system)
    R = 2
    beta = np.full(R, 1.0 / (KB * 120.0))
    p = np.full(R, 30.0 * unit.bar)
    box = np.tile(np.diag(lj.system.getDefaultPeriodicBoxVectors()), (R, 1))
    x = np.stack([lj.positions, lj.positions + 0.002 * np.random.default_rng(1).normal(size=lj.positions.shape)])
    engines = []
    for eng in (cpu_engine_factory(), OracleEngine(ForceFieldOracle)):
        eng.set_system(desc)
        eng.set_states(beta)
        eng.set_integrator('V R O R V', 0.002, 1.0, 12, True, 1e-8)
        eng.set_barostat(p, 5)
        eng.set_energy_const_volume(0.0)
        eng.seed(SEED)
        eng.set_replicas(R, 0, x, None, box, np.arange(R))
        assert not eng.propagate(1).any()
        eng.barostat_attempts(3)
        engines.append(eng)
    cpu, ora = engines
    assert np.allclose(cpu.get_boxes(), ora.get_boxes(), rtol=1e-10)
    assert not np.allclose(cpu.get_boxes(), box)                       # some move was accepted
    assert np.allclose(cpu.get_replicas()[0], ora.get_replicas()[0], atol=1e-8)
    rows_c, rows_o = cpu.compute_energies(), ora.compute_energies()
    assert np.allclose(rows_c, rows_o, rtol=1e-9)
    vs, na, nc = cpu.barostat_stats()
    assert na.tolist() == [5, 5] and (nc <= na).all() and (vs > 0).all()


def test_cpu_fire_minimiser_tracks_the_python_oracle(cpu_engine_factory):
    """remd_minimize on the CPU library = OracleFIRE (integrators.py:2290-2469) step for step: 25 FIRE steps of a jittered
    LJ fluid and of alanine dipeptide with constraints."""
    for system, n_it, jitter in ((ts.LennardJonesFluid(nparticles=216), 25, 0.003), (ts.AlanineDipeptideExplicit(), 8, 0.0)):
        desc = system_to_desc(system.system)
        x = (system.positions + jitter * np.random.default_rng(2).normal(size=system.positions.shape))[None]
        box = np.diag(system.system.getDefaultPeriodicBoxVectors())[None]
        out = []
        for eng in (cpu_engine_factory(), OracleEngine(ForceFieldOracle)):
            eng.set_system(desc)
            eng.set_states(np.array([1.0 / (KB * 300.0)]))
            eng.set_integrator('V R O R V', 0.001, 1.0, 1, True, 1e-8)
            eng.seed(SEED)
            eng.set_replicas(1, 0, x, None, box, np.zeros(1, dtype=np.int64))
            conv, n = eng.minimize(tolerance=0.0, max_iterations=n_it)
            out.append((eng.get_replicas()[0], n))
        assert out[0][1] == out[1][1] == n_it
        assert np.abs(out[0][0] - out[1][0]).max() < 1e-7
        assert np.abs(out[0][0] - x).max() > 1e-5                       # it moved


def test_cpu_library_exports_the_whole_abi(cpu_engine_factory):
    from openmmtools_amd._engine import EXPORTS
    eng = cpu_engine_factory()
    for name in EXPORTS:
        assert hasattr(eng.lib, name), name
    assert eng.lib.remd_cpu_num_threads() >= 1
    with pytest.raises(RuntimeError, match='GPU measurement'):
        eng.roof_microbench()

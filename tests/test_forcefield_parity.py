"""GPU parity of the force-field kernels (csrc/forces.hip, pme.hip, SETTLE/SHAKE in integrate.hip) against the
f64 oracle, through the C ABI.  Tolerances: u_kl / potential 1e-5 relative (north_star); forces to the
reference's own cross-platform bar of 0.06 kcal/mol/A RMSE = 2.5 kJ/mol/nm (scripts/test_openmm_platforms.py:154-155),
in practice ~1e-4 relative; FFT vs numpy to fp32 round-off."""
import numpy as np
import pytest
from openmmtools_amd import testsystems as ts, alchemy
from openmmtools_amd.system import system_to_desc
from oracle import md_oracle as mo
from oracle.forcefield import ForceFieldOracle
from oracle_engine import OracleEngine

pytestmark = pytest.mark.gpu
KB = 0.008314462618153242
SEED = 0xC0FFEE


@pytest.mark.parametrize('shape', [(8, 8, 8), (16, 24, 40), (72, 80, 80), (96, 120, 128), (144, 8, 16)])
def test_fft3d_matches_numpy(hip_engine_factory, shape):
    eng = hip_engine_factory()
    rng = np.random.default_rng(sum(shape))
    a = (rng.normal(size=shape) + 1j * rng.normal(size=shape)).astype(np.complex64)
    ref = np.fft.fftn(a.astype(np.complex128))
    got = eng.test_fft3d(a)
    assert np.abs(got - ref).max() < 2e-5 * np.abs(ref).max()
    back = eng.test_fft3d(got, inverse=True) / np.prod(shape)
    assert np.abs(back - a).max() < 1e-5 * np.abs(a).max()


def _engine_for(eng, system, positions, R=2, temperature=300.0, lam_s=None, jitter=0.0, splitting='V R O R V',
                dt=0.001, n_steps=5, labels=None, econst=None, desc=None):
    desc = system_to_desc(system) if desc is None else desc
    eng.set_system(desc)
    K = R if lam_s is None else len(lam_s)
    eng.set_states(np.full(K, 1.0 / (KB * temperature)), lam_s, None, econst)
    eng.set_integrator(splitting, dt, 1.0, n_steps, True, 1e-8)
    eng.seed(SEED)
    rng = np.random.default_rng(3)
    x = np.stack([positions + jitter * rng.normal(size=positions.shape) * (r > 0) for r in range(R)])
    box = np.tile(np.diag(system.getDefaultPeriodicBoxVectors()), (R, 1))
    eng.set_replicas(R, 0, x, None, box, np.arange(R) % K if labels is None else labels)
    return desc, x, box


def test_lj_fluid_energy_and_forces(hip_engine_factory):
    """BASELINE config 2 system: LennardJonesFluid(512), CutoffPeriodic + switch + dispersion correction."""
    lj = ts.LennardJonesFluid(nparticles=512)
    eng = hip_engine_factory()
    desc, x, box = _engine_for(eng, lj.system, lj.positions, R=2, jitter=0.01)
    ff = ForceFieldOracle(desc)
    rows, U = eng.compute_energies(want_potential=True)
    f = eng.get_forces()
    xd = eng.get_replicas()[0]
    for r in range(2):
        e_ref, f_ref = ff.energy_forces(xd[r], box[r])
        assert np.isclose(U[r], e_ref, rtol=1e-5), (U[r], e_ref)
        assert np.abs(f[r] - f_ref).max() < 2e-4 * np.abs(f_ref).max()
        assert np.isclose(rows[r, 0], e_ref / (KB * 300.0), rtol=1e-5)


@pytest.mark.parametrize('annihilate', [False, True])
def test_lj_fluid_alchemical_ukl(hip_engine_factory, annihilate):
    """16 lambda_sterics states on atoms 0-9 (tests/test_alchemy.py:1864-1866): u_kl rows from one pass; with annihilate_sterics
    the pairs among the ten atoms are lambda-controlled too."""
    lj = ts.LennardJonesFluid(nparticles=512)
    region = alchemy.AlchemicalRegion(alchemical_atoms=range(10), annihilate_sterics=annihilate)
    system = alchemy.AbsoluteAlchemicalFactory().create_alchemical_system(lj.system, region)
    lam = np.linspace(1.0, 0.0, 16)
    nb = [f for f in system.getForces() if hasattr(f, 'particles') and hasattr(f, 'exceptions')][0]
    V = np.prod(np.diag(system.getDefaultPeriodicBoxVectors()))
    econst = alchemy.alchemical_long_range_constants(system, nb, lam, V)
    assert econst[0] < 0 and abs(econst[-1]) < abs(econst[0])       # tail shrinks as the atoms decouple
    eng = hip_engine_factory()
    desc, x, box = _engine_for(eng, system, lj.positions, R=3, lam_s=lam, jitter=0.01, labels=[0, 7, 15],
                               econst=econst)
    ff = ForceFieldOracle(desc)
    rows, U = eng.compute_energies(want_potential=True)
    xd = eng.get_replicas()[0]
    beta = 1.0 / (KB * 300.0)
    for r in range(3):
        ref = beta * (ff.state_energies(xd[r], box[r], lam, np.ones(16)) + econst)
        assert np.allclose(rows[r], ref, rtol=1e-5), np.abs(rows[r] / ref - 1).max()
        # the per-replica potential is the energy in the replica's own state (without the state's long-range constant)
        assert np.isclose(U[r], ff.energy_forces(xd[r], box[r], lambda_sterics=lam[[0, 7, 15][r]], forces=False)[0], rtol=1e-5)
    # forces at each replica's own lambda
    f = eng.get_forces()
    for r, k in enumerate([0, 7, 15]):
        f_ref = ff.energy_forces(xd[r], box[r], lambda_sterics=lam[k])[1]
        assert np.abs(f[r] - f_ref).max() < 2e-4 * np.abs(f_ref).max()


def _lj_alchemical(eng, n_steps, dt, R=3, splitting='V R O R V'):
    lj = ts.LennardJonesFluid(nparticles=512)
    region = alchemy.AlchemicalRegion(alchemical_atoms=range(10))
    system = alchemy.AbsoluteAlchemicalFactory().create_alchemical_system(lj.system, region)
    lam = np.linspace(1.0, 0.0, 16)
    _engine_for(eng, system, lj.positions, R=R, lam_s=lam, jitter=0.01, labels=[0, 7, 15][:R], splitting=splitting, dt=dt, n_steps=n_steps)


@pytest.mark.parametrize('splitting,dt,n_steps', [('V R O R V', 0.001, 60), ('V R R O R R V', 0.001, 600), ('O V R V O', 0.002, 100)])
def test_resident_small_system_kernel_follows_the_regular_launches(hip_engine_factory, monkeypatch, splitting, dt, n_steps):
    """Round 3: systems without constraints, mesh or listed terms of up to 1024 atoms (BASELINE configs 1 and 2) are propagated by
    ONE launch per move (resident_md_kernel: a workgroup per replica, an atom per thread, x / v / f in registers for all n_steps,
    Verlet list in LDS rebuilt when an atom leaves its skin / 2 sphere) instead of ~8 dependent launches per MD step.  Same pair
    arithmetic, same Philox streams: the trajectory follows the regular path to fp32 summation order, through list rebuilds
    (600 steps move the fastest atoms several skins) and for the lambda of each replica's state."""
    out = []
    for flag in ('1', '0'):
        monkeypatch.setenv('REMD_RESIDENT', flag)
        eng = hip_engine_factory()
        _lj_alchemical(eng, n_steps, dt, splitting=splitting)
        assert not eng.propagate(3).any()
        x, v = eng.get_replicas()[:2]
        out.append((x, v, eng.compute_energies()))
    (xa, va, ua), (xb, vb, ub) = out
    moved = np.abs(xb - ts.LennardJonesFluid(nparticles=512).positions[None]).max()
    assert moved > (0.15 if n_steps == 600 else 0.01)                  # the long case outruns the skin: rebuilds happened
    # fp32 summation order differs between the paths; a collision amplifies that (600 steps: the worst atom 4e-4 nm, the median
    # 1e-7 nm), so the bulk is held tightly and the outliers loosely
    dx, dv = np.abs(xa - xb), np.abs(va - vb)
    assert np.median(dx) < 2e-6 and np.median(dv) < 2e-5, (np.median(dx), np.median(dv))
    short = n_steps <= 60
    assert dx.max() < (5e-5 if short else 5e-3), dx.max()
    assert dv.max() < (5e-4 if short else 5e-2), dv.max()
    if short:       # (a pair caught in a collision turns 5e-3 nm into several kT: the long case compares coordinates only)
        assert np.allclose(ua, ub, rtol=1e-4, atol=1e-3)


def test_resident_kernel_list_overflow_falls_back_to_the_regular_launches(hip_engine_factory, monkeypatch):
    """A neighbour list that does not fit the kernel's LDS budget raises the device flag; the propagation is run again from its
    start state by the regular launches and the handle stays on them: bit-identical to a handle that never used the kernel."""
    out = []
    for cap, flag in (('2', '1'), (None, '0')):
        if cap is None:
            monkeypatch.delenv('REMD_RESIDENT_CAP', raising=False)
        else:
            monkeypatch.setenv('REMD_RESIDENT_CAP', cap)
        monkeypatch.setenv('REMD_RESIDENT', flag)
        eng = hip_engine_factory()
        _lj_alchemical(eng, 40, 0.001)
        assert not eng.propagate(0).any()
        assert not eng.propagate(1).any()
        x, v = eng.get_replicas()[:2]
        out.append((x, v))
    assert np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1], out[1][1])


def test_exclusion_word_with_high_lane_bits(hip_engine_factory):
    """8-atom groups whose exclusions include (slot 3, slot 7) of a cluster: bit 31 of the diagonal cluster pair's 64-bit
    exclusion word is set.  (The word is read as two 32-bit halves; a signed low half once smeared that bit over lanes
    32..63 and silently dropped the pairs of i atoms 4..7.)  Also (0, 4) and the chain (k, k+1)."""
    from openmmtools_amd.system import NonbondedForce
    lj = ts.LennardJonesFluid(nparticles=512)
    nb = [f for f in lj.system.getForces() if isinstance(f, NonbondedForce)][0]
    for g in range(64):
        for k in range(7):
            nb.addException(8 * g + k, 8 * g + k + 1, 0.0, 0.1, 0.0)
        nb.addException(8 * g + 3, 8 * g + 7, 0.0, 0.1, 0.0)
        nb.addException(8 * g, 8 * g + 4, 0.0, 0.1, 0.0)
    # every group a compact 2 x 2 x 2 cube (0.38 nm edge), so that its non-excluded pairs (4,6), (4,7), (5,7) ... carry force
    L = float(np.diag(lj.system.getDefaultPeriodicBoxVectors())[0])
    cube = 0.38 * np.array([[a, b, c] for a in (0, 1) for b in (0, 1) for c in (0, 1)], dtype=float)
    centres = (L / 4.0) * np.array([[a, b, c] for a in range(4) for b in range(4) for c in range(4)], dtype=float) + 0.3
    positions = (centres[:, None, :] + cube[None, :, :]).reshape(512, 3)
    eng = hip_engine_factory()
    desc, x, box = _engine_for(eng, lj.system, positions, R=2, jitter=0.01)
    ff = ForceFieldOracle(desc)
    U = eng.compute_energies(want_potential=True)[1]
    f = eng.get_forces()
    xd = eng.get_replicas()[0]
    for r in range(2):
        e_ref, f_ref = ff.energy_forces(xd[r], box[r])
        assert np.isclose(U[r], e_ref, rtol=1e-5), (U[r], e_ref)
        assert np.abs(f[r] - f_ref).max() < 2e-4 * np.abs(f_ref).max()


@pytest.fixture(scope='module')
def alanine():
    al = ts.AlanineDipeptideExplicit()
    return al, system_to_desc(al.system)


def test_alanine_energy_and_forces(hip_engine_factory, alanine):
    """BASELINE config 3 system: bonded + LJ + PME (direct, reciprocal, exclusions, self) + dispersion."""
    al, _ = alanine
    eng = hip_engine_factory()
    desc, x, box = _engine_for(eng, al.system, al.positions, R=2, jitter=0.002)
    ff = ForceFieldOracle(desc)
    rows, U = eng.compute_energies(want_potential=True)
    f = eng.get_forces()
    xd = eng.get_replicas()[0]
    for r in range(2):
        e_ref, f_ref = ff.energy_forces(xd[r], box[r])
        assert np.isclose(U[r], e_ref, rtol=1e-5), (U[r], e_ref, U[r] - e_ref)
        rmse = np.sqrt(((f[r] - f_ref) ** 2).sum(axis=1).mean())
        assert rmse < 2.5, rmse                                    # 0.06 kcal/mol/A in kJ/mol/nm
        assert np.abs(f[r] - f_ref).max() < 1e-3 * np.abs(f_ref).max()


def test_alanine_constraints_and_substeps(hip_engine_factory, alanine):
    """SETTLE (analytic) / SHAKE clusters on the device vs iterative f64 SHAKE/RATTLE in the oracle."""
    al, desc = alanine
    eng, ora = hip_engine_factory(), OracleEngine(ForceFieldOracle)
    _engine_for(eng, al.system, al.positions, R=1, splitting='V R R O R R V', dt=0.002)
    _engine_for(ora, al.system, al.positions, R=1, splitting='V R R O R R V', dt=0.002)
    eng.propagate_zero = None
    # velocities: Maxwell-Boltzmann + velocity constraints
    eng.set_integrator('V R R O R R V', 0.002, 1.0, 0, True, 1e-8)
    ora.set_integrator('V R R O R R V', 0.002, 1.0, 0, True, 1e-8)
    eng.propagate(1)
    ora.propagate(1)
    xg, vg, _, _ = eng.get_replicas()
    assert np.abs(vg - ora.v).max() < 5e-5 * np.abs(ora.v).max() + 1e-5
    cons = ora.sys.constraints
    i, j, dist = np.array([c[0] for c in cons]), np.array([c[1] for c in cons]), np.array([c[2] for c in cons])
    rel_v = np.einsum('ij,ij->i', xg[0][j] - xg[0][i], vg[0][j] - vg[0][i])
    assert np.abs(rel_v).max() < 2e-5
    ora.x, ora.v = xg.copy(), vg.copy()
    for tok in ('R', 'V', 'O', 'R', 'R', 'V'):
        eng.step(tok, iteration=1, first_step=0)
        ora.step(tok, iteration=1, first_step=0)
        xg, vg, _, _ = eng.get_replicas()
        assert np.abs(xg - ora.x).max() < 2e-6, (tok, np.abs(xg - ora.x).max())
        assert np.abs(vg - ora.v).max() < 3e-3 * np.abs(ora.v).max() / 10 + 2e-4, (tok, np.abs(vg - ora.v).max())
        d = np.linalg.norm(xg[0][j] - xg[0][i], axis=1)
        assert np.abs(d - dist).max() < 2e-6, (tok, np.abs(d - dist).max())
        ora.x, ora.v = xg.copy(), vg.copy()


def test_alanine_short_trajectory(hip_engine_factory, alanine):
    """10 g-BAOAB steps at 2 fs with CM-motion removal: device fp32 vs oracle f64, same Philox stream."""
    al, desc = alanine
    eng, ora = hip_engine_factory(), OracleEngine(ForceFieldOracle)
    _engine_for(eng, al.system, al.positions, R=1, splitting='V R R O R R V', dt=0.002, n_steps=10)
    _engine_for(ora, al.system, al.positions, R=1, splitting='V R R O R R V', dt=0.002, n_steps=10)
    assert not eng.propagate(3).any()
    ora.propagate(3)
    xg, vg, _, _ = eng.get_replicas()
    assert np.abs(xg - ora.x).max() < 5e-5                      # nm, after 10 steps
    assert np.sqrt(((vg - ora.v) ** 2).mean()) < 2e-3 * np.sqrt((ora.v ** 2).mean())
    m = np.asarray(desc['mass'])
    p = (m[:, None] * vg[0]).sum(0) / m.sum()
    assert np.abs(p).max() < 5e-3                                 # CM velocity stays ~0 (thermal kicks only)
    U_dev = eng.compute_energies(want_potential=True)[1]
    U_ora = ora.potentials()
    assert np.allclose(U_dev, U_ora, rtol=2e-4)


@pytest.mark.parametrize('grid', [(48, 60, 64), (80, 90, 96), (54, 81, 100), (64, 50, 40), (45, 120, 36), (32, 72, 128), (125, 128, 108),
                                  (64, 64, 64), (64, 64, 45), (128, 128, 128), (128, 128, 50), (48, 64, 64), (60, 128, 128), (64, 64, 128), (128, 128, 64)])
def test_pme_mesh_sizes_through_the_force_path(hip_engine_factory, grid):
    """The LDS-resident mesh passes run mixed-radix stages (radix 4, 2, 3, 5 in that order, pme.hip: factorize) along y and
    x in the plane pass and along z (packed real pairs, nz / 2 points) in the spreading / gathering passes; planes that
    outgrow one workgroup take the slab path (125 x 128); square 64 x 64 and 128 x 128 planes take the register transforms of
    pme_pow2.h (radix 8 x 8 and 16 x 8, with even and odd nz behind them), nz = 64 / 128 with ny a multiple of 64 the register z passes
    (packed 32- and 64-point transforms), also behind planes of other sizes.  Mesh sizes with every radix in a first (no twiddles) and a later
    stage, odd and even nz: energy and forces of the alanine dipeptide box against the f64 oracle on the SAME mesh (the
    mesh only has to be at least as fine as the Ewald tolerance asks, so any of these is a legal choice)."""
    al = ts.AlanineDipeptideExplicit()
    eng = hip_engine_factory()
    desc = system_to_desc(al.system)
    desc['pme_grid'] = np.array(grid, dtype=np.int32)
    eng.set_system(desc)
    eng.set_states(np.full(1, 1.0 / (KB * 300.0)))
    eng.set_integrator('V R O R V', 0.001, 1.0, 1, True, 1e-8)
    box = np.diag(al.system.getDefaultPeriodicBoxVectors())[None]
    eng.set_replicas(1, 0, al.positions[None], None, box, np.arange(1))
    U = eng.compute_energies(want_potential=True)[1]
    f = eng.get_forces()
    e_ref, f_ref = ForceFieldOracle(desc).energy_forces(al.positions, box[0])
    assert np.isclose(U[0], e_ref, rtol=1e-5), (U[0], e_ref)
    assert np.sqrt(((f[0] - f_ref) ** 2).sum(axis=1).mean()) < 0.5


@pytest.mark.parametrize('system_cls', [ts.AlanineDipeptideExplicit, ts.DHFRExplicit])
def test_register_transforms_of_power_of_two_meshes_agree_with_the_scheduled_passes(hip_engine_factory, monkeypatch, system_cls):
    """Round 6: 64^3 (alanine dipeptide, rebalanced split) and 128^3 (DHFR) meshes run their plane pass and their z passes on
    butterflies held in registers (pme_pow2.h, REMD_PME_POW2 bits 0 / 1); the scheduled mixed-radix passes (REMD_PME_POW2=0)
    are another factorisation of the same transforms: energies agree to 1e-7 relative, forces to 2e-4 of the largest force
    component (fp32 round-off of a different summation order), each combination of the two bits."""
    tsys = system_cls()
    desc = system_to_desc(tsys.system, ewald_split='auto')
    assert tuple(desc['pme_grid']) in ((64, 64, 64), (128, 128, 128))
    out = {}
    for mode in ('0', '1', '2', '3'):
        monkeypatch.setenv('REMD_PME_POW2', mode)
        eng = hip_engine_factory()
        _engine_for(eng, tsys.system, tsys.positions, R=1, jitter=0.0, splitting='V R O R V', dt=0.001, n_steps=1, desc=desc)
        out[mode] = (eng.compute_energies(want_potential=True)[1][0], eng.get_forces()[0])
    U0, f0 = out['0']
    for mode in ('1', '2', '3'):
        U, f = out[mode]
        assert abs(U - U0) < 1e-7 * abs(U0), (mode, U, U0)
        assert np.abs(f - f0).max() < 2e-4 * np.abs(f0).max(), (mode, np.abs(f - f0).max())


@pytest.mark.parametrize('system_cls,R', [(ts.HostGuestExplicit, 2), (ts.DHFRExplicit, 1)])
def test_large_systems_energy_forces_and_propagation(hip_engine_factory, system_cls, R):
    """BASELINE config 4 / 5 systems (CB7:B2 host-guest, 4491 atoms, 96^3 mesh; DHFR, 23558 atoms, 144^3 mesh):
    energy within 1e-5 relative, force RMSE within the cross-platform bar, and a short g-BAOAB run keeps every
    constraint and stays finite."""
    tsys = system_cls()
    eng = hip_engine_factory()
    desc, x, box = _engine_for(eng, tsys.system, tsys.positions, R=R, jitter=0.001, splitting='V R R O R R V',
                               dt=0.002, n_steps=10)
    ff = ForceFieldOracle(desc)
    U = eng.compute_energies(want_potential=True)[1]
    f = eng.get_forces()
    xd = eng.get_replicas()[0]
    for r in range(R):
        e_ref, f_ref = ff.energy_forces(xd[r], box[r])
        assert np.isclose(U[r], e_ref, rtol=1e-5), (U[r], e_ref)
        rmse = np.sqrt(((f[r] - f_ref) ** 2).sum(axis=1).mean())
        assert rmse < 2.5, rmse
        assert np.abs(f[r] - f_ref).max() < 2e-3 * np.abs(f_ref).max(), np.abs(f[r] - f_ref).max()      # per atom, not only on average
    assert not eng.propagate(1).any()
    xg, vg, _, _ = eng.get_replicas()
    cons = mo.OracleSystem(desc).constraints
    i, j, dist = np.array([c[0] for c in cons]), np.array([c[1] for c in cons]), np.array([c[2] for c in cons])
    for r in range(R):
        d = np.linalg.norm(xg[r][j] - xg[r][i], axis=1)
        assert np.abs(d - dist).max() < 5e-6
    assert np.isfinite(eng.compute_energies()).all()


def test_hostguest_alchemical_electrostatics_and_sterics_ukl(hip_engine_factory):
    """BASELINE config 4 state family on CB7:B2 (alchemical guest = atoms 126-155, tests/test_alchemy.py:1873-1875):
    lambda_electrostatics 1 -> 0 with exact PME treatment (charge offsets, alchemy.py:1897-1899), then lambda_sterics
    1 -> 0 (soft core).  The device fits U(lambda_e) = a + b l + c l^2 from three energy passes; the oracle
    recomputes every state from scratch."""
    hg = ts.HostGuestExplicit()
    region = alchemy.AlchemicalRegion(alchemical_atoms=range(126, 156))
    system = alchemy.AbsoluteAlchemicalFactory().create_alchemical_system(hg.system, region)
    lam_e = np.array([1.0, 0.5, 0.25, 0.0, 0.0, 0.0, 0.0])
    lam_s = np.array([1.0, 1.0, 1.0, 1.0, 0.6, 0.2, 0.0])
    K = len(lam_e)
    eng = hip_engine_factory()
    desc = system_to_desc(system)
    eng.set_system(desc)
    beta = 1.0 / (KB * 300.0)
    eng.set_states(np.full(K, beta), lam_s, lam_e, None)
    eng.set_integrator('V R R O R R V', 0.002, 1.0, 5, True, 1e-8)
    eng.seed(SEED)
    box = np.tile(np.diag(system.getDefaultPeriodicBoxVectors()), (2, 1))
    x = np.stack([hg.positions, hg.positions + 0.001 * np.random.default_rng(0).normal(size=hg.positions.shape)])
    labels = np.array([1, 5])
    eng.set_replicas(2, 0, x, None, box, labels)
    rows = eng.compute_energies()
    xd = eng.get_replicas()[0]
    ff = ForceFieldOracle(desc)
    for r in range(2):
        ref = beta * ff.state_energies(xd[r], box[r], lam_s, lam_e)
        assert np.allclose(rows[r], ref, rtol=1e-5), np.abs(rows[r] / ref - 1).max()
        assert abs((rows[r, 0] - rows[r, 3]) - (ref[0] - ref[3])) < 0.05 * K      # differences survive to << 1 kT
    f = eng.get_forces()
    for r, k in enumerate(labels):
        f_ref = ff.energy_forces(xd[r], box[r], lambda_sterics=lam_s[k], lambda_electrostatics=lam_e[k])[1]
        assert np.sqrt(((f[r] - f_ref) ** 2).sum(axis=1).mean()) < 2.5
    assert not eng.propagate(1).any()


def test_alanine_propagation_is_bit_reproducible(hip_engine_factory):
    """Forces are accumulated as 64-bit fixed point (integer atomics), the mesh charges as 32-bit fixed point, energies
    in fixed-order partial sums, and every random stream is counter based: two runs of the same propagate on two streams
    must agree bit for bit (positions, velocities, u_kl), whatever the order in which workgroups and atomics retire."""
    al = ts.AlanineDipeptideExplicit()
    out = []
    for _ in range(2):
        eng = hip_engine_factory()
        _engine_for(eng, al.system, al.positions, R=3, jitter=0.002, splitting='V R R O R R V', dt=0.002, n_steps=25)
        assert not eng.propagate(4).any()
        x, v, _, _ = eng.get_replicas()
        out.append((x, v, eng.compute_energies()))
    assert np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1], out[1][1])
    assert np.array_equal(out[0][2], out[1][2])


def test_bins_from_the_integrator_chain_do_not_change_a_bit(hip_engine_factory, monkeypatch):
    """Inside remd_run_steps the integrator chain that finalises the positions also bins the atoms by PME mesh column
    (REMD_PME_CHAINBIN, default on), replacing the binning launch; the order of atoms inside a bin differs from run to run,
    charges and forces are fixed-point sums: bit-identical trajectories either way."""
    al = ts.AlanineDipeptideExplicit()
    out = []
    for flag, merge in (('1', '1'), ('0', '1'), ('1', '0'), ('0', '0')):
        monkeypatch.setenv('REMD_PME_CHAINBIN', flag)
        monkeypatch.setenv('REMD_CHAIN_MERGE', merge)     # centre-of-mass sum by a barrier inside ONE chain launch, or two launches
        eng = hip_engine_factory()
        _engine_for(eng, al.system, al.positions, R=3, jitter=0.002, splitting='V R R O R R V', dt=0.002, n_steps=30)
        eng.propagate(0)
        eng.propagate(1)
        x, v = eng.get_replicas()[:2]
        out.append((x, v))
    for x, v in out[1:]:
        assert np.array_equal(x, out[0][0]) and np.array_equal(v, out[0][1])


def test_overfull_mesh_bins_fall_back_to_the_binning_launch(hip_engine_factory, monkeypatch):
    """the bins the integrator chain fills for the PME passes are capped (4 x the mean occupancy of a mesh column); a
    density the cap does not hold (slab, droplet, vacuum layer) used to leave a sticky error flag and a dead handle.  Now the
    propagation is run again from its start state with the binning launch (no cap) and the handle stays on it: same
    trajectory, bit for bit, as a run that never used the capped bins.  The overflow is provoked with a cap of 8 atoms."""
    al = ts.AlanineDipeptideExplicit()
    out = []
    for cap, chainbin in (('8', '1'), (None, '0')):
        if cap is None:
            monkeypatch.delenv('REMD_PME_CBIN_CAP', raising=False)
        else:
            monkeypatch.setenv('REMD_PME_CBIN_CAP', cap)
        monkeypatch.setenv('REMD_PME_CHAINBIN', chainbin)
        eng = hip_engine_factory()
        _engine_for(eng, al.system, al.positions, R=3, jitter=0.002, splitting='V R R O R R V', dt=0.002, n_steps=30)
        assert not eng.propagate(0).any()              # first call: overflow -> recovery inside remd_propagate
        assert not eng.propagate(1).any()              # the handle keeps working
        x, v = eng.get_replicas()[:2]
        out.append((x, v, eng.compute_energies()))
    assert np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1], out[1][1])
    assert np.array_equal(out[0][2], out[1][2])


def test_resident_pair_workgroups_do_not_change_a_bit(hip_engine_factory, monkeypatch):
    """The pair kernel either launches one workgroup per work item or keeps a resident set that pulls items from a queue
    (REMD_NB_PERSIST_GRID, chosen by timing at run time): forces are fixed-point sums, so positions and velocities after a
    propagation are bit-identical under every choice, and so are the forces themselves."""
    al = ts.AlanineDipeptideExplicit()
    out = []
    for grid in ('0', '96', '512'):
        monkeypatch.setenv('REMD_NB_PERSIST_GRID', grid)
        eng = hip_engine_factory()
        _engine_for(eng, al.system, al.positions, R=3, jitter=0.002, splitting='V R R O R R V', dt=0.002, n_steps=25)
        f = eng.get_forces()
        eng.propagate(0)
        x, v = eng.get_replicas()[:2]
        out.append((f, x, v))
    for f, x, v in out[1:]:
        assert np.array_equal(f, out[0][0]) and np.array_equal(x, out[0][1]) and np.array_equal(v, out[0][2])


def test_stream_modes_of_a_force_evaluation_do_not_change_a_bit(hip_engine_factory, monkeypatch):
    """Round 4: which stream is treated as the critical one is a run-time choice (tuner candidate '0p'): the pair kernel at raised
    wave priority, the listed terms on the mesh stream -- as a launch of their own or as extra workgroups of the spreading launch --
    and the join by the scatter's per-replica done counters instead of a signal launch; round 6: the listed terms one ATOM per thread
    (the default; REMD_LISTED_ATOMS=0: one term per thread, 6 - 12 atomics each), and the mesh forces summed by position in the bins
    and handed to the atoms once (REMD_PME_FBIN=0: five scattered triples per atom).  Every contribution is an integer atomic
    add into the same accumulators, so forces, positions and velocities are bit-identical under every combination, at the
    rebalanced Ewald split as well."""
    al = ts.AlanineDipeptideExplicit()
    desc = system_to_desc(al.system, ewald_split='auto')
    modes = [dict(REMD_NB_PRIO='0', REMD_NB_PERSIST_GRID='0'),
             dict(REMD_NB_PRIO='1', REMD_NB_PERSIST_GRID='0'),
             dict(REMD_NB_PRIO='1', REMD_NB_PERSIST_GRID='0', REMD_LISTED_RIDE='0'),
             dict(REMD_NB_PRIO='1', REMD_NB_PERSIST_GRID='0', REMD_NB_FOLD='0'),
             dict(REMD_NB_PRIO='1', REMD_NB_PERSIST_GRID='0', REMD_LISTED_MAIN='0'),
             dict(REMD_NB_PRIO='1', REMD_NB_PERSIST_GRID='0', REMD_LISTED_ATOMS='0'),
             dict(REMD_NB_PRIO='1', REMD_NB_PERSIST_GRID='0', REMD_LISTED_ATOMS='0', REMD_LISTED_RIDE='0'),
             dict(REMD_NB_PRIO='0', REMD_NB_PERSIST_GRID='0', REMD_LISTED_ATOMS='0'),
             dict(REMD_NB_PRIO='1', REMD_NB_PERSIST_GRID='0', REMD_PME_FBIN='0'),
             dict(REMD_NB_PRIO='0', REMD_NB_PERSIST_GRID='0', REMD_PME_FBIN='0', REMD_LISTED_ATOMS='0'),
             dict(REMD_NB_PRIO='1', REMD_NB_PERSIST_GRID='640')]
    out = []
    for mode in modes:
        for k in ('REMD_NB_PRIO', 'REMD_NB_PERSIST_GRID', 'REMD_LISTED_RIDE', 'REMD_NB_FOLD', 'REMD_LISTED_MAIN', 'REMD_LISTED_ATOMS', 'REMD_PME_FBIN'):
            monkeypatch.delenv(k, raising=False)
        for k, v in mode.items():
            monkeypatch.setenv(k, v)
        eng = hip_engine_factory()
        _engine_for(eng, al.system, al.positions, R=3, jitter=0.002, splitting='V R R O R R V', dt=0.002, n_steps=25, desc=desc)
        f = eng.get_forces()
        eng.propagate(0)
        x, v = eng.get_replicas()[:2]
        out.append((f, x, v))
    for f, x, v in out[1:]:
        assert np.array_equal(f, out[0][0]) and np.array_equal(x, out[0][1]) and np.array_equal(v, out[0][2])


def test_phases_of_a_propagation_do_not_change_a_bit(hip_engine_factory):
    """Round 6: remd_propagate runs a handle's replicas as two blocks whose MD steps take turns on the device (remd_set_phases): the
    integrator chain of one block beside the pair and mesh kernels of the other.  Replicas are independent between mixes
    (multistatesampler.py:1296-1297), forces are fixed-point sums, noise is keyed by the global replica, the schedule of spatial
    re-sorts restarts with every propagation, and the first kick uses the forces the energy pass left in both modes: positions,
    velocities and the energy matrix after two iterations (with an energy pass and new labels in between, 60 steps each: one
    re-sort inside) are identical to the one-block run, for an even and a ragged split."""
    al = ts.AlanineDipeptideExplicit()
    desc = system_to_desc(al.system, ewald_split='auto')
    out = []
    for phases, R in ((1, 5), (2, 5), (1, 4), (2, 4)):
        eng = hip_engine_factory()
        eng.set_phases(phases)
        _engine_for(eng, al.system, al.positions, R=R, jitter=0.002, splitting='V R R O R R V', dt=0.002, n_steps=60, desc=desc)
        eng.propagate(0)
        u0 = eng.compute_energies()
        eng.set_labels(np.arange(R)[::-1].copy())
        eng.propagate(1)
        assert eng.phases_active() == phases
        x, v = eng.get_replicas()[:2]
        out.append((x, v, u0, eng.compute_energies()))
    for a, b in ((0, 1), (2, 3)):
        for q in range(4):
            assert np.array_equal(out[a][q], out[b][q]), (a, b, q)


def test_handles_propagated_concurrently_equal_handles_propagated_one_after_the_other(hip_engine_factory):
    """profiles/r05_11_group_overlap.txt reported other positions for the 24 replicas as 2 or 3 concurrently propagated
    handles than as one.  On this tree the same experiment is bit-identical in every mode (profiles/r06_1_concurrent_handles.txt:
    sequential, two host threads, flags or events, tuner pinned or not); what can differ is a handle whose device-side poll ran out --
    its propagation is run again from the snapshot with a fresh spatial order (different fp32 summation order), which is what the
    round-5 tree did under the load of several polling handles.  Here: two handles from two host threads, the same two one after the
    other, and remd_propagate_many (one thread, steps taking turns) give the same bits."""
    import threading
    from openmmtools_amd._engine import HipEngine
    al = ts.AlanineDipeptideExplicit()
    desc = system_to_desc(al.system, ewald_split='auto')
    box = np.diag(al.system.getDefaultPeriodicBoxVectors())
    R = 6
    x0 = np.tile(al.positions, (R, 1, 1)) + np.random.default_rng(7).normal(0, 0.002, (R,) + al.positions.shape)
    beta = 1.0 / (KB * np.geomspace(300.0, 400.0, R))
    res = {}
    for mode in ('sequential', 'threads', 'many'):
        engs = []
        for a, b in ((0, 3), (3, 6)):
            e = hip_engine_factory()
            e.set_phases(1)
            e.set_system(desc); e.set_states(beta)
            e.set_integrator('V R R O R R V', 0.002, 1.0, 50, True, 1e-8)
            e.seed(SEED)
            e.set_replicas(R, a, x0[a:b], None, np.tile(box, (b - a, 1)), np.arange(R))
            engs.append(e)
        for it in range(2):
            if mode == 'sequential':
                for e in engs: e.propagate(it)
            elif mode == 'threads':
                th = [threading.Thread(target=e.propagate, args=(it,)) for e in engs]
                for t in th: t.start()
                for t in th: t.join()
            else:
                HipEngine.propagate_many(engs, it)
        res[mode] = [np.concatenate([e.get_replicas()[q] for e in engs]) for q in (0, 1)]
    for mode in ('threads', 'many'):
        assert np.array_equal(res[mode][0], res['sequential'][0]) and np.array_equal(res[mode][1], res['sequential'][1]), mode


@pytest.mark.parametrize('system_cls', [ts.AlanineDipeptideExplicit, ts.HostGuestExplicit])
def test_cluster_pair_lists_match_the_tile_kernel(hip_engine_factory, monkeypatch, system_cls):
    """The direct-space sum runs on per-tile union lists with every cluster pair listed once (Newton's third law, sci
    kernel); a list that outgrows its capacity falls back to the 64-atom tile kernel, which walks every tile pair from both
    sides (REMD_NB_TILES forces it).  Same pairs, same per-pair arithmetic: forces agree to the order of summation
    (fixed-point accumulation, fp32 partial sums), energies to the fp32 per-lane partial sums of the tile kernel."""
    tsys = system_cls()
    res = []
    for tiles in (False, True):
        if tiles:
            monkeypatch.setenv('REMD_NB_TILES', '1')
        eng = hip_engine_factory()
        _engine_for(eng, tsys.system, tsys.positions, R=2, jitter=0.002)
        U = eng.compute_energies(want_potential=True)[1]
        res.append((eng.get_forces(), U))
    (f1, u1), (f0, u0) = res
    assert np.abs(f1 - f0).max() < 2e-5 * np.abs(f0).max()
    assert np.allclose(u1, u0, rtol=1e-7, atol=1e-6)          # (fp32 partial sums per lane in both kernels, summed in f64)


def test_more_than_8191_molecules_stay_on_the_cluster_path(hip_engine_factory, monkeypatch):
    """Round 4: the molecule sort kept its work arrays in LDS, which capped the cluster-pair path at 8191 molecules (larger
    systems fell to the tile kernel).  Past that the same ranking runs on a global scratch buffer with 64-bit keys
    (sort_groups_large_kernel).  10 000 single-atom molecules (a dense LJ fluid): the cluster path against the oracle and
    against the tile kernel on the same coordinates."""
    lj = ts.LennardJonesFluid(nparticles=10000, reduced_density=0.5)
    L = np.diag(lj.system.getDefaultPeriodicBoxVectors())
    g = np.stack(np.meshgrid(*[np.arange(22)] * 3, indexing='ij'), axis=-1).reshape(-1, 3)
    pos = (np.random.default_rng(5).permutation(g)[:10000] + 0.5) * (L / 22)          # jittered lattice sites, in no spatial order
    res = []
    for tiles in (False, True):
        if tiles:
            monkeypatch.setenv('REMD_NB_TILES', '1')
        eng = hip_engine_factory()
        desc, x, box = _engine_for(eng, lj.system, pos, R=2, jitter=0.02)
        U = eng.compute_energies(want_potential=True)[1]
        res.append((eng.get_forces(), U, eng.get_replicas()[0]))
    (f1, u1, xd), (f0, u0, _) = res
    assert np.abs(f1 - f0).max() < 2e-5 * np.abs(f0).max()
    assert np.allclose(u1, u0, rtol=2e-7)
    ff = ForceFieldOracle(desc)
    e_ref, f_ref = ff.energy_forces(xd[1], box[1])
    assert np.isclose(u1[1], e_ref, rtol=1e-5), (u1[1], e_ref)
    assert np.abs(f1[1] - f_ref).max() < 2e-4 * np.abs(f_ref).max()


def test_softcore_exceptions_of_a_region_that_cuts_a_molecule(hip_engine_factory):
    """Round 4: the Lennard-Jones part of a 1-4 exception between an alchemical and a non-alchemical atom is soft-core and
    lambda_sterics-controlled (the factory's CustomBondForce, alchemy.py:1836-1851, 1985-1998).  Alanine dipeptide with its
    first 10 atoms alchemical (16 such exceptions): u_kl over a (lambda_e, lambda_s) ladder, the own-state potential and the
    forces at each replica's lambda against the f64 oracle."""
    al = ts.AlanineDipeptideExplicit()
    system = alchemy.AbsoluteAlchemicalFactory().create_alchemical_system(al.system, alchemy.AlchemicalRegion(alchemical_atoms=range(10)))
    lam_e = np.array([1.0, 0.5, 0.0, 0.0, 0.0, 0.0])
    lam_s = np.array([1.0, 1.0, 1.0, 0.7, 0.3, 0.0])
    nb = [f for f in system.getForces() if hasattr(f, 'particles') and hasattr(f, 'exceptions')][0]
    V = np.prod(np.diag(system.getDefaultPeriodicBoxVectors()))
    econst = alchemy.alchemical_long_range_constants(system, nb, lam_s, V)
    eng = hip_engine_factory()
    desc = system_to_desc(system)
    eng.set_system(desc)
    beta = 1.0 / (KB * 300.0)
    eng.set_states(np.full(6, beta), lam_s, lam_e, econst)
    eng.set_integrator('V R R O R R V', 0.002, 1.0, 5, True, 1e-8)
    eng.seed(SEED)
    labels = np.array([0, 3, 4])
    x = np.stack([al.positions + 0.001 * r * np.random.default_rng(r).normal(size=al.positions.shape) for r in range(3)])
    box = np.tile(np.diag(system.getDefaultPeriodicBoxVectors()), (3, 1))
    eng.set_replicas(3, 0, x, None, box, labels)
    ff = ForceFieldOracle(desc)
    rows, U = eng.compute_energies(want_potential=True)
    xd = eng.get_replicas()[0]
    f = eng.get_forces()
    for r, k in enumerate(labels):
        ref = ff.state_energies(xd[r], box[r], lam_s, lam_e)
        assert np.allclose(rows[r], beta * (ref + econst), rtol=1e-5), np.abs(rows[r] / (beta * (ref + econst)) - 1).max()
        assert np.isclose(U[r], ref[k], rtol=1e-5)
        f_ref = ff.energy_forces(xd[r], box[r], lambda_sterics=lam_s[k], lambda_electrostatics=lam_e[k])[1]
        assert np.abs(f[r] - f_ref).max() < 2e-4 * np.abs(f_ref).max()


def test_annihilated_sterics(hip_engine_factory):
    """AlchemicalRegion(annihilate_sterics=True) (alchemy.py:421, 1767-1779, 1841-1846; remd_set_alchemical_options): the
    Lennard-Jones pairs and 1-4 exceptions inside the alchemical region are soft-core and lambda_sterics-controlled like those
    with the environment.  CB7:B2 with the guest annihilated: u_kl over a (lambda_e, lambda_s) ladder, the own-state potential and
    the forces at each replica's lambda against the f64 oracle; the same ladder under decoupling differs by tens of kJ/mol."""
    hg = ts.HostGuestExplicit()
    system = alchemy.AbsoluteAlchemicalFactory().create_alchemical_system(
        hg.system, alchemy.AlchemicalRegion(alchemical_atoms=range(126, 156), annihilate_sterics=True))
    lam_e = np.array([1.0, 0.5, 0.0, 0.0, 0.0, 0.0])
    lam_s = np.array([1.0, 1.0, 1.0, 0.7, 0.3, 0.0])
    nb = [f for f in system.getForces() if hasattr(f, 'particles') and hasattr(f, 'exceptions')][0]
    V = np.prod(np.diag(system.getDefaultPeriodicBoxVectors()))
    econst = alchemy.alchemical_long_range_constants(system, nb, lam_s, V)
    eng = hip_engine_factory()
    desc = system_to_desc(system)
    assert desc['annihilate_sterics'] is True
    eng.set_system(desc)
    beta = 1.0 / (KB * 300.0)
    eng.set_states(np.full(6, beta), lam_s, lam_e, econst)
    eng.set_integrator('V R R O R R V', 0.002, 1.0, 5, True, 1e-8)
    eng.seed(SEED)
    labels = np.array([1, 3, 5])
    x = np.stack([hg.positions + 0.001 * r * np.random.default_rng(r).normal(size=hg.positions.shape) for r in range(3)])
    box = np.tile(np.diag(system.getDefaultPeriodicBoxVectors()), (3, 1))
    eng.set_replicas(3, 0, x, None, box, labels)
    ff = ForceFieldOracle(desc)
    rows, U = eng.compute_energies(want_potential=True)
    xd = eng.get_replicas()[0]
    f = eng.get_forces()
    for r, k in enumerate(labels):
        ref = ff.state_energies(xd[r], box[r], lam_s, lam_e)
        assert np.allclose(rows[r], beta * (ref + econst), rtol=1e-5), np.abs(rows[r] / (beta * (ref + econst)) - 1).max()
        assert np.isclose(U[r], ref[k], rtol=1e-5)
        f_ref = ff.energy_forces(xd[r], box[r], lambda_sterics=lam_s[k], lambda_electrostatics=lam_e[k])[1]
        assert np.abs(f[r] - f_ref).max() < 2e-4 * np.abs(f_ref).max()
    decoupled = ForceFieldOracle(system_to_desc(alchemy.AbsoluteAlchemicalFactory().create_alchemical_system(
        hg.system, alchemy.AlchemicalRegion(alchemical_atoms=range(126, 156)))))
    assert abs(ff.state_energies(xd[0], box[0], lam_s, lam_e)[5] - decoupled.state_energies(xd[0], box[0], lam_s, lam_e)[5]) > 10.0


def test_config4_all_64_alchemical_states_ukl(hip_engine_factory):
    """BASELINE config 4 AT ITS STATED SIZE: CB7:B2 with the full 64-state ladder (lambda_electrostatics 1 -> 0 over 32
    states, then lambda_sterics 1 -> 0 over 32) — every column of two replicas' u_kl rows against
    the oracle, which recomputes each state from scratch (reduced_potential_at_states, states.py:911-992).  The device
    fits U(lambda_e) from three mesh passes and evaluates the soft-core pairs for all lambda_s in one pass; this is the
    check that the fit and the one-pass evaluation hold on all 64 states, not on a 7-state sample."""
    hg = ts.HostGuestExplicit()
    region = alchemy.AlchemicalRegion(alchemical_atoms=range(126, 156))
    system = alchemy.AbsoluteAlchemicalFactory().create_alchemical_system(hg.system, region)
    lam_e = np.concatenate([np.linspace(1.0, 0.0, 32), np.zeros(32)])
    lam_s = np.concatenate([np.ones(32), np.linspace(1.0, 0.0, 32)])
    K = 64
    nb = [f for f in system.getForces() if hasattr(f, 'particles') and hasattr(f, 'exceptions')][0]
    V = np.prod(np.diag(system.getDefaultPeriodicBoxVectors()))
    econst = alchemy.alchemical_long_range_constants(system, nb, lam_s, V)
    eng = hip_engine_factory()
    desc = system_to_desc(system)
    eng.set_system(desc)
    beta = 1.0 / (KB * 300.0)
    eng.set_states(np.full(K, beta), lam_s, lam_e, econst)
    eng.set_integrator('V R R O R R V', 0.002, 1.0, 5, True, 1e-8)
    eng.seed(SEED)
    box = np.tile(np.diag(system.getDefaultPeriodicBoxVectors()), (2, 1))
    x = np.stack([hg.positions, hg.positions + 0.001 * np.random.default_rng(0).normal(size=hg.positions.shape)])
    labels = np.array([10, 50])
    eng.set_replicas(2, 0, x, None, box, labels)
    rows = eng.compute_energies()
    assert rows.shape == (2, 64)
    xd = eng.get_replicas()[0]
    ff = ForceFieldOracle(desc)
    for r in range(2):
        ref = beta * (ff.state_energies(xd[r], box[r], lam_s, lam_e) + econst)
        assert np.allclose(rows[r], ref, rtol=1e-5), np.abs(rows[r] / ref - 1).max()
        # what mixing consumes are differences between neighbouring states: they must survive the cancellation
        d_dev, d_ref = np.diff(rows[r]), np.diff(ref)
        assert np.abs(d_dev - d_ref).max() < 0.05, np.abs(d_dev - d_ref).max()


def test_config4_alchemical_ukl_at_the_rebalanced_ewald_split(hip_engine_factory):
    """Round 4: what HipEngine asks for by default on config 4 (Coulomb range 1.091 nm, 80 x 80 x 80 instead of 90 x 90 x 90; the
    soft-core and Lennard-Jones terms keep the 1.0 nm cutoff): a 12-state sample of the 64-state ladder against the f64 oracle at
    the SAME split, and the device rows against the device rows at OpenMM's split (the Ewald sum does not depend on the split: both
    sides' Ewald tolerance bounds the difference), incl. the exact-PME lambda_electrostatics polynomial and forces."""
    hg = ts.HostGuestExplicit()
    region = alchemy.AlchemicalRegion(alchemical_atoms=range(126, 156))
    system = alchemy.AbsoluteAlchemicalFactory().create_alchemical_system(hg.system, region)
    lam_e = np.concatenate([np.linspace(1.0, 0.0, 6), np.zeros(6)])
    lam_s = np.concatenate([np.ones(6), np.linspace(1.0, 0.0, 6)])
    K = 12
    nb = [f for f in system.getForces() if hasattr(f, 'particles') and hasattr(f, 'exceptions')][0]
    V = np.prod(np.diag(system.getDefaultPeriodicBoxVectors()))
    econst = alchemy.alchemical_long_range_constants(system, nb, lam_s, V)
    beta = 1.0 / (KB * 300.0)
    box = np.tile(np.diag(system.getDefaultPeriodicBoxVectors()), (2, 1))
    x = np.stack([hg.positions, hg.positions + 0.001 * np.random.default_rng(0).normal(size=hg.positions.shape)])
    rows, forces = {}, {}
    for split in ('reference', 'auto'):
        desc = system_to_desc(system, ewald_split=split)
        assert (max(desc['pme_grid']) == 80 and desc['coulomb_cutoff'] > 1.05) if split == 'auto' else 'coulomb_cutoff' not in desc
        eng = hip_engine_factory()
        eng.set_system(desc)
        eng.set_states(np.full(K, beta), lam_s, lam_e, econst)
        eng.set_integrator('V R R O R R V', 0.002, 1.0, 5, True, 1e-8)
        eng.seed(SEED)
        eng.set_replicas(2, 0, x, None, box, np.array([2, 9]))
        rows[split] = eng.compute_energies()
        forces[split] = eng.get_forces()
        if split == 'auto':
            xd = eng.get_replicas()[0]
            ff = ForceFieldOracle(desc)
            for r in range(2):
                ref = beta * (ff.state_energies(xd[r], box[r], lam_s, lam_e) + econst)
                assert np.allclose(rows[split][r], ref, rtol=1e-5), np.abs(rows[split][r] / ref - 1).max()
                assert np.abs(np.diff(rows[split][r]) - np.diff(ref)).max() < 0.05
    assert np.allclose(rows['auto'], rows['reference'], rtol=1e-5), np.abs(rows['auto'] / rows['reference'] - 1).max()
    rmse = np.sqrt(((forces['auto'] - forces['reference']) ** 2).sum(axis=2).mean())
    assert rmse < 0.5, rmse                                    # kJ/mol/nm; the reference's cross-platform bar is 25.1


def test_config5_dhfr_128_state_sams_row_and_jump(hip_engine_factory):
    """BASELINE config 5 AT ITS STATED SHAPE: DHFR (23 558 atoms) with a 128-state ladder (temperatures, BASELINE.md
    section 4 leaves the ladder to us: geomspace 300-400 K) and the SAMS global jump.  Two replicas: the 128-column u_kl
    rows against the oracle's potential, then one remd_mix(SAMS) on the device matrix against the C oracle's jump
    (which the reference-executed fixture pins) on the same rows — labels and count matrices bit-exact."""
    import oracle
    dh = ts.DHFRExplicit()
    K = 128
    T = np.geomspace(300.0, 400.0, K)
    beta = 1.0 / (KB * T)
    eng = hip_engine_factory()
    desc = system_to_desc(dh.system)
    eng.set_system(desc)
    eng.set_states(beta, None, None, None)
    eng.set_integrator('V R R O R R V', 0.002, 1.0, 5, True, 1e-8)
    eng.seed(SEED)
    R = 2
    box = np.tile(np.diag(dh.system.getDefaultPeriodicBoxVectors()), (R, 1))
    x = np.stack([dh.positions, dh.positions + 0.001 * np.random.default_rng(1).normal(size=dh.positions.shape)])
    labels = np.array([3, 100], dtype=np.int64)
    eng.set_replicas(R, 0, x, None, box, labels)
    rows, U = eng.compute_energies(want_potential=True)
    assert rows.shape == (R, K)
    ff = ForceFieldOracle(desc)
    xd = eng.get_replicas()[0]
    for r in range(R):
        e_ref = ff.potential(xd[r], box[r])
        assert np.isclose(U[r], e_ref, rtol=1e-5), (U[r], e_ref)
        assert np.allclose(rows[r], beta * e_ref, rtol=1e-5)
        assert np.allclose(rows[r], beta * U[r], rtol=1e-14)        # paralleltempering.py:206-215 outer product
    # the jump: energies this large make the categorical draw degenerate unless the weights compensate (as SAMS's
    # adapted weights do): log_w = +u of a reference replica, so that log P spans a few kT across the ladder
    logw = rows[0] + np.linspace(0.0, 3.0, K)
    got = eng.mix('sams-global-jump', 7, labels, R=R, K=K, log_weights=logw)
    ref = oracle.mix('sams-global-jump', SEED, 7, rows, labels, log_weights=logw)
    assert np.array_equal(got[0], ref[0])
    assert np.array_equal(got[1], ref[1]) and np.array_equal(got[2], ref[2])
    assert np.allclose(got[3], ref[3], rtol=0, atol=1e-9)      # log P of O(1e5)-sized arguments: 1e-9 absolute


def test_config5_dhfr_hamiltonian_ladder_columns(hip_engine_factory):
    """BASELINE config 5 as north_star words it -- DHFR with 128 HAMILTONIAN replicas: an alchemical ladder (ten solvent
    molecules decoupled: lambda_electrostatics 1 -> 0 over 64 states, then lambda_sterics 1 -> 0 over 64) on the 23 558-atom
    system.  The device fits U(lambda_e) from three mesh passes and evaluates the soft-core pairs of all lambda_s in one pass;
    five columns spread over the ladder against the oracle, which recomputes each state from scratch
    (reduced_potential_at_states, states.py:911-992), and neighbouring columns differ by a few kT."""
    dh = ts.DHFRExplicit()
    n = dh.system.getNumParticles()
    region = alchemy.AlchemicalRegion(alchemical_atoms=range(n - 30, n))          # the last ten waters
    system = alchemy.AbsoluteAlchemicalFactory().create_alchemical_system(dh.system, region)
    K = 128
    lam_e = np.concatenate([np.linspace(1.0, 0.0, 64), np.zeros(64)])
    lam_s = np.concatenate([np.ones(64), np.linspace(1.0, 0.0, 64)])
    nb = [f for f in system.getForces() if hasattr(f, 'particles') and hasattr(f, 'exceptions')][0]
    V = np.prod(np.diag(system.getDefaultPeriodicBoxVectors()))
    econst = alchemy.alchemical_long_range_constants(system, nb, lam_s, V)
    eng = hip_engine_factory()
    desc = system_to_desc(system)
    eng.set_system(desc)
    beta = 1.0 / (KB * 300.0)
    eng.set_states(np.full(K, beta), lam_s, lam_e, econst)
    eng.set_integrator('V R R O R R V', 0.002, 1.0, 5, True, 1e-8)
    eng.seed(SEED)
    box = np.diag(system.getDefaultPeriodicBoxVectors())[None]
    eng.set_replicas(1, 0, dh.positions[None], None, box, np.array([40], dtype=np.int64))
    rows = eng.compute_energies()
    assert rows.shape == (1, K) and np.isfinite(rows).all()
    ff = ForceFieldOracle(desc)
    xd = eng.get_replicas()[0][0]
    for k in (0, 31, 63, 96, 127):
        e_ref = ff.energy_forces(xd, box[0], lambda_sterics=lam_s[k], lambda_electrostatics=lam_e[k], forces=False)[0] + econst[k]
        assert np.isclose(rows[0, k], beta * e_ref, rtol=1e-5), (k, rows[0, k], beta * e_ref)
    assert np.abs(np.diff(rows[0])).max() < 10.0                 # ten waters discharged over 64 states: 6.3 kT per rung


def test_constraint_tolerance_sets_the_newton_iterations_of_the_xh_solve(hip_engine_factory):
    """Round 6: the X-H position solve iterates until every bond of a cluster is within constraint_tolerance (relative, as OpenMM's
    addConstrainPositions that integrators.py:1416-1418 calls), no tighter than 2e-7 and with at most 8 updates, instead of a fixed
    three.  A loose tolerance stops after fewer updates than a tight one; anything below the fp32 floor is the floor, bit for bit;
    no solve runs into the bound; bond lengths read back from the stored (absolute, fp32) coordinates are within the tolerance or
    the ~1e-5 that storing 0.1 nm bonds at 3 nm from the origin costs."""
    al = ts.AlanineDipeptideExplicit()
    cons = mo.OracleSystem(system_to_desc(al.system)).constraints
    res = {}
    for tol in (1e-3, 2e-7, 1e-8):
        eng = hip_engine_factory()
        _engine_for(eng, al.system, al.positions, R=2, jitter=0.002, splitting='V R R O R R V', dt=0.002, n_steps=60)
        eng.set_integrator('V R R O R R V', 0.002, 1.0, 60, True, tol)
        eng.propagate(0)
        x, v = eng.get_replicas()[:2]
        i, j, d = np.array([c[0] for c in cons]), np.array([c[1] for c in cons]), np.array([c[2] for c in cons])
        rel = np.abs(np.linalg.norm(x[:, i] - x[:, j], axis=-1) - d) / d
        res[tol] = (x, v, rel.max(), eng.constraint_stats())
    for tol, (x, v, worst, (n_it, unconverged)) in res.items():
        assert not unconverged and 1 <= n_it <= 8, (tol, n_it, unconverged)
        assert worst < max(1.5 * tol, 2e-5), (tol, worst)
    assert res[1e-3][3][0] < res[1e-8][3][0], (res[1e-3][3], res[1e-8][3])          # fewer updates at the loose tolerance
    assert np.array_equal(res[2e-7][0], res[1e-8][0]) and np.array_equal(res[2e-7][1], res[1e-8][1])


def test_energy_drift_bounds_the_constraint_solver(hip_engine_factory):
    """The device solves X-H clusters by Newton iterations to constraint_tolerance (no tighter than 2e-7: the fp32 floor) and rigid
    waters analytically (include/remd_hip.h).  What that means for the dynamics, in the terms that matter: velocity
    Verlet ('V R V', no thermostat) on alanine dipeptide in water at 1 fs conserves K + U to a small fraction of kT per
    degree of freedom over 0.4 ps -- the bar OpenMM's own constraint tolerance (1e-5 by default there) is held to."""
    al = ts.AlanineDipeptideExplicit()
    eng = hip_engine_factory()
    _engine_for(eng, al.system, al.positions, R=1, splitting='V R V', dt=0.001, n_steps=100)
    eng.minimize(tolerance=50.0, max_iterations=200)
    eng.set_integrator('V R O R V', 0.001, 5.0, 200, True, 1e-8)
    eng.propagate(0)                                                  # thermalise for 0.2 ps
    eng.set_integrator('V R V', 0.001, 0.0, 100, False, 1e-8)
    energies = []
    for it in range(5):
        kinetic = eng.get_replicas(positions=False, velocities=False, kinetic=True)[3]
        energies.append(float(kinetic[0] + eng.compute_energies(want_potential=True)[1][0]))
        if it < 4:
            eng.propagate(it + 1)
    ndof = 3 * 2269 - 2259 - 3
    drift = (np.array(energies) - energies[0]) / (ndof * KB * 300.0)
    assert np.abs(drift).max() < 2e-3, drift                          # kT per degree of freedom over 0.4 ps

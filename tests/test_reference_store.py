"""Reading and resuming a store WRITTEN BY THE REFERENCE (the half that was missing: the reference's reader
could not be replaced because this package only understood its own record files).

Fixture: tests/golden/reference_store/alanine_dipeptide_legacy{,_checkpoint}.nc, the netCDF4 store the reference ships for its
own resume test (openmmtools/tests/test_sampling.py:2943-2990; copied by tests/golden/make_golden_reference_store.py).  The test
below follows that reference test line by line: no 'velocities' variable in the checkpoint file; from_storage restores the sampler;
every sampler state has zero velocities; the simulation is extended by one iteration; it can be loaded again.  What the reference
does with openmm + netCDF4, this package does through ctypes on libhdf5 (openmmtools_amd/multistate/_hdf5.py) and the device
engine -- on the CPU here (libremd_cpu.so through the same C ABI), on the GPU under -m gpu."""
import os

import numpy as np
import pytest

import oracle
from openmmtools_amd import states
from openmmtools_amd.multistate import MultiStateReporter, MultiStateSampler, ReplicaExchangeSampler
from openmmtools_amd.multistate import _hdf5
from openmmtools_amd._engine import HipEngine

HERE = os.path.dirname(os.path.abspath(__file__))
STORE = os.path.join(HERE, 'golden', 'reference_store', 'alanine_dipeptide_legacy.nc')
CHECKPOINT = os.path.join(HERE, 'golden', 'reference_store', 'alanine_dipeptide_legacy_checkpoint.nc')
CPU_LIB = os.path.join(os.path.dirname(os.path.abspath(oracle.__file__)), '_build', 'libremd_cpu.so')
KB = 0.008314462618153242

pytestmark = pytest.mark.skipif(not _hdf5.available(), reason='no libhdf5 on this machine')


def test_reference_store_is_read_like_the_references_reporter_reads_it():
    rep = MultiStateReporter(STORE, open_mode='r')
    assert rep.is_reference_store and rep.storage_exists()
    assert rep.checkpoint_interval == 1                                   # global attribute CheckpointInterval
    assert rep.read_last_iteration(last_checkpoint=False) == 2 and rep.read_last_iteration() == 2
    assert rep.read_checkpoint_iterations() == [0, 1, 2]
    thermo, unsampled = rep.read_thermodynamic_states()                   # multistatereporter.py:552-610
    assert len(thermo) == 20 and unsampled == []
    assert thermo[0].temperature == 300.0 and abs(thermo[-1].temperature - 600.0) < 1e-9
    assert all(t.system is thermo[0].system for t in thermo)              # '_Reporter__compatible_state': one System object
    assert thermo[0].n_particles == 2269 and thermo[0].pressure is None and thermo[0].system.getNumConstraints() == 2259
    moves = rep.read_mcmc_moves()                                         # :795-811
    assert len(moves) == 20 and type(moves[0]).__name__ == 'LangevinSplittingDynamicsMove'
    m = moves[0]
    assert (m.timestep, m.collision_rate, m.n_steps, m.splitting, m.reassign_velocities) == (0.004, 5.0, 10, 'V R O R V', False)
    assert m.n_restart_attempts == 20 and m.constraint_tolerance == 1e-6
    opts = rep.read_dict('options')
    assert opts['number_of_iterations'] == 2 and opts['locality'] is None and opts['online_analysis_interval'] == 200
    assert 'yank.multistate' in rep.read_dict('metadata')['title']
    e, nb, eu = rep.read_energies()
    assert e.shape == (3, 1, 20) and nb.shape == (3, 1, 20) and eu.shape == (3, 1, 0) and np.all(nb == 1)
    assert np.array_equal(rep.read_replica_thermodynamic_states(), np.zeros((3, 1), np.int64))
    acc, prop = rep.read_mixing_statistics()
    assert acc.shape == (3, 20, 20) and prop.sum() == 0                   # a MultiStateSampler does not mix
    ss = rep.read_sampler_states(2)
    assert len(ss) == 1 and ss[0].positions.shape == (2269, 3) and np.all(ss[0].velocities == 0)
    assert np.allclose(np.diag(ss[0].box_vectors), [3.2852862, 3.2861648, 3.1855097], atol=1e-6)
    # the stored energies ARE beta_l U(x) of the stored frame (the OpenMM-fixture tests pin the value; here: consistency of
    # what is read): one replica => u[l] T_l is constant
    T = np.array([t.temperature for t in thermo])
    assert np.abs((e[2, 0] * T) / (e[2, 0, 0] * T[0]) - 1).max() < 1e-12
    with pytest.raises(IOError):
        rep.write_last_iteration(3)                                       # the reference's file is never written
    with pytest.raises(IOError):
        MultiStateReporter(STORE, open_mode='w')


def _resume_and_extend(engine, tmp_path):
    """openmmtools/tests/test_sampling.py:2943-2990, restated on this package's classes."""
    # "Assert no velocities in legacy dataset variables"
    with _hdf5.File(CHECKPOINT) as ck:
        assert 'velocities' not in ck.keys('/')[1]
    # "Load repex simulation"
    sampler = MultiStateSampler.from_storage(STORE, engine=engine, continue_in=str(tmp_path / 'continued.nc'))
    assert sampler.iteration == 2 and sampler.n_replicas == 1 and sampler.n_states == 20
    # "Assert velocities are initialized as zeros"
    for state in sampler.sampler_states:
        assert np.all(state.velocities == 0), 'velocities in sampler state from legacy checkpoint are expected to be all zeros'
    e_before = sampler.energy_thermodynamic_states.copy()
    # "Resume simulation"
    sampler.extend(n_iterations=1)
    assert sampler.iteration == 3 and sampler.number_of_iterations == 3
    x = sampler.sampler_states[0].positions
    assert np.isfinite(x).all() and np.isfinite(sampler.energy_thermodynamic_states).all()
    assert not np.array_equal(sampler.energy_thermodynamic_states, e_before)
    # 10 steps of 4 fs from REST (zero velocities, no reassignment): potential energy flows into the 4548 degrees of freedom,
    # u drops by a few hundred kT of the ~2270 kT equipartition would take, and stays the same order
    assert -0.15 < sampler.energy_thermodynamic_states[0, 0] / e_before[0, 0] - 1.0 < 0.15
    assert sampler.energy_thermodynamic_states[0, 0] < e_before[0, 0]
    # "delete reporters and load again": the continuation ('.nc' => again the reference's layout, written by this package) holds
    # the whole history
    sampler._reporter.close()
    del sampler
    rep = MultiStateReporter(str(tmp_path / 'continued.nc'), open_mode='r')
    assert rep.is_reference_store and rep.read_last_iteration() == 3
    with _hdf5.File(str(tmp_path / 'continued_checkpoint.nc')) as ck:
        assert 'velocities' in ck.keys('/')[1]                               # stores written today carry them (:1795-1806)
    e, nb, eu = rep.read_energies()
    ref = MultiStateReporter(STORE, open_mode='r')
    assert e.shape == (4, 1, 20) and np.array_equal(e[:3], ref.read_energies()[0])
    rep.close()
    again = MultiStateSampler.from_storage(str(tmp_path / 'continued.nc'), engine=engine.spawn() if hasattr(engine, 'spawn') else engine)
    assert again.iteration == 3
    v = again.sampler_states[0].velocities
    assert v is not None and np.abs(v).max() > 0                            # velocities are part of a checkpoint now
    return e


def test_resume_velocities_from_legacy_storage_cpu_library(tmp_path):
    if not os.path.exists(CPU_LIB):
        oracle.build()
    eng = HipEngine(lib_path=CPU_LIB)
    eng.is_device = False
    try:
        _resume_and_extend(eng, tmp_path)
    finally:
        eng.close()


def test_a_mixing_sampler_class_resumes_the_same_store(tmp_path):
    """The calling class decides what is built (the reference: cls(**options)): a ReplicaExchangeSampler needs one replica per
    state and refuses this one-replica store with the reference's own message."""
    with pytest.raises((ValueError, RuntimeError)):
        ReplicaExchangeSampler.from_storage(STORE, engine=None)


@pytest.mark.gpu
def test_resume_velocities_from_legacy_storage_on_the_device(hip_engine_factory, tmp_path):
    e = _resume_and_extend(hip_engine_factory(), tmp_path)
    assert np.isfinite(e).all()


# ---- writing the same layout -----------------------------------------------------------------------------------------

def _h5_structure(path):
    """{object path: (datatype, dataspace, [dimension names], {attribute names})} from `h5dump -H -A`."""
    import re
    import shutil
    import subprocess
    h5dump = shutil.which('h5dump') or '/opt/conda/bin/h5dump'
    if not os.path.exists(h5dump):
        pytest.skip('no h5dump on this machine')
    text = subprocess.run([h5dump, '-H', '-A', path], capture_output=True, text=True, check=True).stdout
    out, stack, cur, in_attr, attr_depth = {}, [], None, None, 0
    lines = text.splitlines()
    depth_of = []
    for i, line in enumerate(lines):
        s = line.strip()
        m = re.match(r'(GROUP|DATASET) "([^"]*)" \{', s)
        if m and in_attr is None:
            name = m.group(2)
            stack.append(name if name != '/' else '')
            depth_of.append(line.index(s[0]))
            if m.group(1) == 'DATASET':
                cur = '/'.join(stack)
                out[cur] = dict(type=None, space=None, dims=[], attrs=set())
            continue
        m = re.match(r'ATTRIBUTE "([^"]*)" \{', s)
        if m and in_attr is None:
            in_attr, attr_depth = m.group(1), line.index(s[0])
            owner = '/'.join(stack)
            out.setdefault(owner or '/', dict(type=None, space=None, dims=[], attrs=set()))['attrs'].add(in_attr)
            continue
        if in_attr is not None:
            if in_attr == 'DIMENSION_LIST':
                out['/'.join(stack)]['dims'] += re.findall(r'DATASET \d+ (/\S+)', s)
            if s == '}' and line.index('}') == attr_depth:
                in_attr = None
            continue
        if s.startswith('DATATYPE') and cur == '/'.join(stack) and out[cur]['type'] is None:
            extra = [x.strip() for x in lines[i + 1:i + 6] if 'STRSIZE' in x or 'CSET' in x] if 'H5T_STRING' in s else []
            extra = [x if 'STRSIZE' not in x or 'VARIABLE' in x else 'STRSIZE n;' for x in extra]      # (fixed lengths differ by content)
            out[cur]['type'] = ' '.join((s + ' ' + ' '.join(extra)).split())
        if s.startswith('DATASPACE') and cur == '/'.join(stack) and out[cur]['space'] is None:
            out[cur]['space'] = s
        if s == '}' and stack and line.index('}') == depth_of[-1]:
            stack.pop()
            depth_of.pop()
            cur = None
    return out


def _pt_run(tmp_path, n_iterations, name='run.nc', interval=2, metadata=None):
    import sys
    sys.path.insert(0, HERE)
    from oracle_engine import OracleEngine
    from openmmtools_amd import testsystems, mcmc, unit
    from openmmtools_amd.multistate import ParallelTemperingSampler
    ho = testsystems.HarmonicOscillator()
    ts = states.ThermodynamicState(ho.system, 300.0 * unit.kelvin)
    ss = states.SamplerState(ho.positions + 0.01, box_vectors=ho.system.getDefaultPeriodicBoxVectors())
    move = mcmc.LangevinSplittingDynamicsMove(timestep=2.0 * unit.femtosecond, collision_rate=5.0 / unit.picosecond, n_steps=10,
                                              reassign_velocities=True, splitting='V R O R V')
    s = ParallelTemperingSampler(mcmc_moves=move, number_of_iterations=n_iterations, engine=OracleEngine(), seed=5)
    rep = MultiStateReporter(str(tmp_path / name), checkpoint_interval=interval)
    s.create(ts, [ss], storage=rep, min_temperature=300.0, max_temperature=400.0, n_temperatures=3,
             unsampled_thermodynamic_states=[states.ThermodynamicState(ho.system, T * unit.kelvin) for T in (250.0, 500.0)],
             metadata=metadata)
    return s, rep


def test_a_run_reported_into_an_nc_path_is_a_store_of_the_references_layout(tmp_path):
    """storage = '<name>.nc': the analysis file and '<name>_checkpoint.nc' are netCDF-4 (HDF5) files with the reference's
    variables (multistatereporter.py:476-1115, 1597-1737); everything read back through the reader that reads the reference's own
    files equals what the sampler held."""
    s, rep = _pt_run(tmp_path, 4, metadata=dict(title='oscillators on a ladder', phase=dict(name='test', index=3)))
    history = []
    for _ in range(4):
        s.run(1)
        history.append((s.energy_thermodynamic_states.copy(), s._energy_unsampled_states.copy(), s.replica_thermodynamic_states.copy(),
                        s._n_accepted_matrix.copy(), s._n_proposed_matrix.copy(), np.stack([st.positions for st in s.sampler_states])))
    assert sorted(os.listdir(tmp_path)) == ['run.nc', 'run_checkpoint.nc']
    r = MultiStateReporter(str(tmp_path / 'run.nc'), open_mode='r')
    assert r.is_reference_store and r.checkpoint_interval == 2
    assert r.read_last_iteration(last_checkpoint=False) == 4 and r.read_checkpoint_iterations() == [0, 2, 4]
    e, nb, eu = r.read_energies()
    st = r.read_replica_thermodynamic_states()
    acc, prop = r.read_mixing_statistics()
    assert e.shape == (5, 3, 3) and eu.shape == (5, 3, 2) and nb.dtype == np.int8 and nb.all()
    for it, (E, EU, L, A, P, X) in enumerate(history, start=1):
        assert np.array_equal(e[it], E) and np.array_equal(eu[it], EU) and np.array_equal(st[it], L)
        assert np.array_equal(acc[it], A) and np.array_equal(prop[it], P)
        if it % 2 == 0:
            got = np.stack([q.positions for q in r.read_sampler_states(it)])
            assert np.array_equal(got, X.astype(np.float32).astype(np.float64))          # f4 on disk (:1621-1632)
    assert r.read_sampler_states(3) is None
    thermo, unsampled = r.read_thermodynamic_states()
    assert [round(t.temperature, 6) for t in thermo] == [300.0, round(np.sqrt(300.0 * 400.0), 6), 400.0] and [u.temperature for u in unsampled] == [250.0, 500.0]
    assert thermo[1].system is thermo[0].system and unsampled[0].system is thermo[0].system      # '_Reporter__compatible_state'
    opts = r.read_dict('options')                     # what the reference's from_storage passes to cls(**options) (:948-950)
    assert opts == dict(locality=None, number_of_iterations=4, online_analysis_interval=200, online_analysis_minimum_iterations=200,
                        online_analysis_target_error=0.0, replica_mixing_scheme='swap-all')
    assert r._ref.read_seed() == 5 and r.read_mcmc_moves()[0].n_steps == 10 and len(r.read_timestamp()) == 5
    # :1127-1139: the title is the file's global attribute, the rest of the metadata is stored nested
    assert r.read_dict('metadata') == dict(title='oscillators on a ladder', phase=dict(name='test', index=3))
    assert r.read_dict('metadata/phase/index') == 3


def test_written_store_has_the_object_structure_of_a_store_the_reference_wrote(tmp_path):
    """Same HDF5 objects as netCDF4-python produced for the reference (h5dump -H of the shipped legacy store): dimension
    scales (IEEE_F32BE, CLASS / NAME / _Netcdf4Dimid), the record dimension unlimited, every variable with the same type, the
    same attached dimensions in the same order and the same kind of attributes; string variables as the reference stores them
    (fixed-length characters for the states, variable-length UTF-8 for options / moves / timestamps)."""
    s, rep = _pt_run(tmp_path, 2, interval=1)
    s.run()
    rep.close()
    ours, theirs = _h5_structure(str(tmp_path / 'run.nc')), _h5_structure(STORE)
    for dim in ('/scalar', '/iteration', '/spatial', '/replica', '/state'):
        assert ours[dim]['type'] == theirs[dim]['type'] == 'DATATYPE H5T_IEEE_F32BE'
        assert {'CLASS', 'NAME', '_Netcdf4Dimid'} <= ours[dim]['attrs'] and {'CLASS', 'NAME', '_Netcdf4Dimid'} <= theirs[dim]['attrs']
    assert 'H5S_UNLIMITED' in ours['/iteration']['space'] and 'H5S_UNLIMITED' in theirs['/iteration']['space']
    for var in ('/energies', '/neighborhoods', '/states', '/accepted', '/proposed', '/timestamp', '/last_iteration', '/options', '/metadata',
                '/thermodynamic_states/state0', '/thermodynamic_states/state1', '/mcmc_moves/move0'):
        a, b = ours[var], theirs[var]
        assert a['type'] == b['type'], (var, a['type'], b['type'])
        da = [d if not d.startswith('/fixedL') else '/fixedL' for d in a['dims']]
        db = [d if not d.startswith('/fixedL') else '/fixedL' for d in b['dims']]
        assert da == db, (var, a['dims'], b['dims'])
        assert {'DIMENSION_LIST', '_Netcdf4Coordinates'} <= a['attrs']
        assert b['attrs'] - {'_Netcdf4Coordinates'} <= a['attrs'], (var, b['attrs'] - a['attrs'])
    for attr in ('Conventions', 'ConventionVersion', 'DataUsedFor', 'CheckpointInterval', 'UUID', 'title', 'program', '_NCProperties'):
        assert attr in ours['/']['attrs'] and attr in theirs['/']['attrs']
    ck_ours, ck_theirs = _h5_structure(str(tmp_path / 'run_checkpoint.nc')), _h5_structure(CHECKPOINT)
    for var in ('/positions', '/box_vectors', '/volumes'):
        assert ck_ours[var]['type'] == ck_theirs[var]['type'] and ck_ours[var]['dims'] == ck_theirs[var]['dims'], var
    assert '/velocities' in ck_ours                       # (the shipped file predates 0.21.3 and has none, :1795-1806)


def test_resume_in_place_continues_the_same_run(tmp_path):
    a, _ = _pt_run(tmp_path, 6, name='a.nc')
    a.run()
    b, repb = _pt_run(tmp_path, 4, name='b.nc')
    b.run()
    repb.close()
    del b
    import sys
    sys.path.insert(0, HERE)
    from oracle_engine import OracleEngine
    from openmmtools_amd.multistate import ParallelTemperingSampler
    r = ParallelTemperingSampler.from_storage(str(tmp_path / 'b.nc'), engine=OracleEngine())
    assert r.iteration == 4 and r._seed == 5
    r.extend(2)
    assert r.iteration == 6 and list(r.replica_thermodynamic_states) == list(a.replica_thermodynamic_states)
    ea = MultiStateReporter(str(tmp_path / 'a.nc'), open_mode='r').read_energies()[0]
    r._reporter.close()
    eb = MultiStateReporter(str(tmp_path / 'b.nc'), open_mode='r').read_energies()[0]
    assert ea.shape == eb.shape == (7, 3, 3)
    assert np.array_equal(ea[:5], eb[:5]) and np.allclose(ea[5:], eb[5:], rtol=2e-5, atol=1e-6)     # restart from f4 checkpoints


def test_the_references_own_files_are_never_written(tmp_path):
    import shutil
    shutil.copy(STORE, tmp_path / 'ref.nc')
    shutil.copy(CHECKPOINT, tmp_path / 'ref_checkpoint.nc')
    before = open(tmp_path / 'ref.nc', 'rb').read()
    with pytest.raises(IOError, match='written by the reference'):
        MultiStateReporter(str(tmp_path / 'ref.nc'), open_mode='w')
    rep = MultiStateReporter(str(tmp_path / 'ref.nc'), open_mode='a')          # opens, read-only
    with pytest.raises(IOError, match='read-only'):
        rep.write_last_iteration(7)
    rep.close()
    assert open(tmp_path / 'ref.nc', 'rb').read() == before


def test_sams_stage_bookkeeping_survives_a_resume_from_the_nc_layout(tmp_path):
    """sams.py:374-393, 615-620: logZ, stage and t0 live under online_analysis/ (latest value and per-iteration history)."""
    import sys
    sys.path.insert(0, HERE)
    from oracle_engine import OracleEngine
    from openmmtools_amd import testsystems, mcmc, unit
    from openmmtools_amd.multistate import SAMSSampler
    ho = testsystems.HarmonicOscillator()
    sts = [states.ThermodynamicState(ho.system, T * unit.kelvin) for T in np.linspace(300.0, 500.0, 5)]
    ss = states.SamplerState(ho.positions, box_vectors=ho.system.getDefaultPeriodicBoxVectors())
    move = mcmc.LangevinSplittingDynamicsMove(timestep=2.0 * unit.femtosecond, collision_rate=5.0 / unit.picosecond, n_steps=10,
                                              reassign_velocities=True, splitting='V R O R V')

    def make(name, n):
        s = SAMSSampler(mcmc_moves=move, number_of_iterations=n, engine=OracleEngine(), seed=3, flatness_criteria='minimum-visits')
        s.create(sts, [ss] * 2, storage=MultiStateReporter(str(tmp_path / name), checkpoint_interval=2))
        return s
    a = make('a.nc', 8)
    a.run()
    b = make('b.nc', 4)
    b.run()
    b._reporter.close()
    r = SAMSSampler.from_storage(str(tmp_path / 'b.nc'), engine=OracleEngine())
    assert r.iteration == 4 and r._stage == b._stage and r._t0 == b._t0 and np.array_equal(r._logZ, b._logZ)
    assert np.array_equal(r._cached_state_histogram, b._cached_state_histogram)
    r.extend(4)
    assert list(r.replica_thermodynamic_states) == list(a.replica_thermodynamic_states)
    assert np.allclose(r._logZ, a._logZ, rtol=1e-6, atol=1e-6)             # (positions restart from f4 checkpoints)
    with _hdf5.File(str(tmp_path / 'b.nc')) as f:
        assert set(f.keys('/online_analysis')[1]) >= {'logZ', 'logZ_history', 'log_weights_history', 'stage', 't0'}
        assert f.read('/online_analysis/logZ_history').shape == (9, 5)


def test_analyzer_works_on_the_nc_layout(tmp_path):
    """The analyzer reads energies / states / options through the reporter interface: the free energy between the ladder's end
    states from a store in the reference's layout equals the one from the record container of the same run, and sits within
    6 sigma of -3/2 ln(T_hi / T_lo)."""
    from openmmtools_amd.multistate import analysis as an
    out = []
    for name in ('run.nc', 'run_records'):
        s, rep = _pt_run(tmp_path, 150, name=name, interval=50)
        s.run()
        D, dD = an.MultiStateSamplerAnalyzer(rep).get_free_energy()
        out.append((D, dD))
    (Da, dDa), (Db, dDb) = out
    assert np.allclose(Da, Db, rtol=0, atol=1e-12) and np.allclose(dDa, dDb, rtol=0, atol=1e-12)
    exact = -1.5 * np.log(500.0 / 250.0)                 # between the two unsampled end states (multistateanalyzer.py:1517-1536)
    assert Da.shape == (5, 5) and abs(Da[0, -1] - exact) < 6.0 * dDa[0, -1] + 0.05


def test_moves_the_layout_cannot_hold_fall_back_to_the_record_container(tmp_path, caplog):
    """What this package does not write of the netCDF4 layout (moves other than the Langevin ones; round 4 added compound
    alchemical states, tests/test_alchemical_store_cpu.py): the same '.nc' path then becomes a record-file container (a
    directory), with a warning; layout='records' asks for that explicitly."""
    import logging
    import sys
    sys.path.insert(0, HERE)
    from oracle_engine import OracleEngine
    from oracle.forcefield import ForceFieldOracle
    from openmmtools_amd import testsystems, mcmc, unit
    lj = testsystems.LennardJonesFluid(nparticles=64)
    ths = [states.ThermodynamicState(lj.system, t * unit.kelvin) for t in (120.0, 130.0)]
    ss = states.SamplerState(lj.positions, box_vectors=lj.system.getDefaultPeriodicBoxVectors())
    move = mcmc.GHMCMove(timestep=1.0 * unit.femtosecond, n_steps=2)
    s = ReplicaExchangeSampler(mcmc_moves=move, number_of_iterations=1, engine=OracleEngine(ForceFieldOracle), seed=1)
    with caplog.at_level(logging.WARNING):
        s.create(ths, [ss], storage=str(tmp_path / 'alch.nc'))
    assert any('GHMCMove' in r.getMessage() and 'record-file container' in r.getMessage() for r in caplog.records)
    s.run()
    assert os.path.isdir(tmp_path / 'alch.nc') and os.path.exists(tmp_path / 'alch.nc' / 'meta.json')
    r = MultiStateReporter(str(tmp_path / 'alch.nc'), open_mode='r')
    assert not r.is_reference_store and r.read_energies()[0].shape == (2, 2, 2)
    forced = MultiStateReporter(str(tmp_path / 'plain.nc'), open_mode='w', layout='records')
    assert os.path.isdir(tmp_path / 'plain.nc') and not forced.is_reference_store


def test_analysis_particles_are_stored_every_iteration_in_the_analysis_file(tmp_path):
    """multistatereporter.py:369-388, 722-741: 'analysis_particle_indices' is a variable of every store (the reference's open()
    would otherwise try to create it, which fails on a file opened for reading); the flagged particles' positions and velocities go
    to the analysis file at EVERY iteration, the full frames to the checkpoint file on the checkpoint interval only."""
    import sys
    sys.path.insert(0, HERE)
    from oracle_engine import OracleEngine
    from oracle.forcefield import ForceFieldOracle
    from openmmtools_amd import testsystems, mcmc, unit
    from openmmtools_amd.multistate import ParallelTemperingSampler
    lj = testsystems.LennardJonesFluid(nparticles=27)
    ts = states.ThermodynamicState(lj.system, 120.0 * unit.kelvin)
    ss = states.SamplerState(lj.positions, box_vectors=lj.system.getDefaultPeriodicBoxVectors())
    move = mcmc.LangevinSplittingDynamicsMove(timestep=1.0 * unit.femtosecond, n_steps=3, reassign_velocities=True, splitting='V R O R V')
    rep = MultiStateReporter(str(tmp_path / 'p.nc'), checkpoint_interval=2, analysis_particle_indices=(3, 5, 11))
    s = ParallelTemperingSampler(mcmc_moves=move, number_of_iterations=3, engine=OracleEngine(ForceFieldOracle), seed=2)
    s.create(ts, [ss], storage=rep, min_temperature=120.0, max_temperature=150.0, n_temperatures=2)
    s.run()
    x = np.stack([st.positions for st in s.sampler_states])
    rep.close()
    with _hdf5.File(str(tmp_path / 'p.nc')) as f:
        assert list(f.read('/analysis_particle_indices')) == [3, 5, 11]
        assert f.read('/positions').shape == (4, 2, 3, 3) and f.read('/velocities').shape == (4, 2, 3, 3)
    with _hdf5.File(str(tmp_path / 'p_checkpoint.nc')) as f:
        assert f.read('/positions').shape == (2, 2, 27, 3)                      # iterations 0 and 2
    r = MultiStateReporter(str(tmp_path / 'p.nc'), open_mode='r')
    sub = r.read_sampler_states(3, analysis_particles_only=True)
    assert np.array_equal(np.stack([q.positions for q in sub]), x[:, [3, 5, 11]].astype(np.float32).astype(np.float64))
    assert r.read_sampler_states(3) is None
    plain = MultiStateReporter(str(tmp_path / 'run.nc'), open_mode='w')
    plain.close()
    with _hdf5.File(str(tmp_path / 'run.nc')) as f:                             # none flagged: the variable exists, empty
        assert f.shape('/analysis_particle_indices') == (0,)


def test_a_checkpoint_file_of_another_simulation_is_refused(tmp_path):
    """tests/test_sampling.py:2143-2171 / multistatereporter.py:318-354: the analysis and checkpoint files of one store carry one
    UUID; pairing an analysis file with the checkpoint file of another run raises IOError, for reading and for appending."""
    from openmmtools_amd.multistate import _hdf5
    sa, ra = _pt_run(tmp_path, 2, name='a.nc')
    sa.run()
    sb, rb = _pt_run(tmp_path, 2, name='b.nc')
    sb.run()
    ra.close(), rb.close()
    with _hdf5.File(str(tmp_path / 'a.nc')) as fa, _hdf5.File(str(tmp_path / 'a_checkpoint.nc')) as fc, \
            _hdf5.File(str(tmp_path / 'b_checkpoint.nc')) as fo:
        uid = lambda f: str(np.asarray(f.attr('UUID')).reshape(-1)[0])
        assert uid(fa) == uid(fc) != uid(fo)
    ok = MultiStateReporter(str(tmp_path / 'a.nc'), open_mode='r')
    assert ok.read_last_iteration(last_checkpoint=False) == 2
    ok.close()
    for mode in ('r', 'a'):
        with pytest.raises(IOError, match='Checkpoint UUID does not match analysis UUID'):
            MultiStateReporter(str(tmp_path / 'a.nc'), checkpoint_storage=str(tmp_path / 'b_checkpoint.nc'), open_mode=mode)


@pytest.mark.parametrize('position_interval,velocity_interval', [(2, 2), (1, 0)])
def test_position_and_velocity_intervals_of_the_analysis_trajectory(tmp_path, position_interval, velocity_interval):
    """tests/test_sampling.py:700-775 (multistatereporter.py:1686-1692): the flagged particles' positions / velocities reach the
    analysis file only every position_interval / velocity_interval iterations (0: never); skipped frames read as zeros;
    checkpoints always carry everything."""
    from openmmtools_amd.multistate import _hdf5
    rng = np.random.default_rng(4)
    rep = MultiStateReporter(str(tmp_path / 't.nc'), open_mode='w', checkpoint_interval=2, analysis_particle_indices=(1, 2),
                             position_interval=position_interval, velocity_interval=velocity_interval)
    assert (rep.position_interval, rep.velocity_interval) == (position_interval, velocity_interval)
    sampler_states = [states.SamplerState(rng.normal(size=(5, 3)), velocities=rng.normal(size=(5, 3))) for _ in range(2)]
    for it in range(3):
        rep.write_sampler_states(sampler_states, it)
        rep.write_last_iteration(it)
    rep.close()
    with _hdf5.File(str(tmp_path / 't.nc')) as f:
        assert int(np.asarray(f.attr('PositionInterval')).reshape(-1)[0]) == position_interval
        assert int(np.asarray(f.attr('VelocityInterval')).reshape(-1)[0]) == velocity_interval
    r = MultiStateReporter(str(tmp_path / 't.nc'), open_mode='r')
    f4 = lambda a: np.asarray(a, dtype=np.float32).astype(np.float64)
    for it in range(3):
        got = r.read_sampler_states(it, analysis_particles_only=True)
        for st, back in zip(sampler_states, got):
            want_x = f4(st.positions[[1, 2]]) if position_interval and it % position_interval == 0 else np.zeros((2, 3))
            want_v = f4(st.velocities[[1, 2]]) if velocity_interval and it % velocity_interval == 0 else np.zeros((2, 3))
            assert np.array_equal(back.positions, want_x), (it, back.positions)
            assert np.array_equal(back.velocities, want_v), it
    full = r.read_sampler_states(2)                                        # the checkpoint frame: all particles, both arrays
    assert np.array_equal(full[1].positions, f4(sampler_states[1].positions)) and np.array_equal(full[1].velocities, f4(sampler_states[1].velocities))


def test_stored_analysis_particles_take_priority_over_the_argument(tmp_path):
    """tests/test_sampling.py:816-866."""
    blank = str(tmp_path / 'temp_dir' / 'blank_analysis.nc')
    MultiStateReporter(blank, open_mode='w', analysis_particle_indices=()).close()
    assert MultiStateReporter(blank, open_mode='r', analysis_particle_indices=(0, 1)).analysis_particle_indices == ()
    set1 = str(tmp_path / 'temp_dir' / 'set1_analysis.nc')
    MultiStateReporter(set1, open_mode='w', analysis_particle_indices=(0, 1)).close()
    assert MultiStateReporter(set1, open_mode='r', analysis_particle_indices=()).analysis_particle_indices == (0, 1)
    assert MultiStateReporter(set1, open_mode='r', analysis_particle_indices=(0, 2)).analysis_particle_indices == (0, 1)
    assert MultiStateReporter(str(tmp_path / 'unopened.nc'), analysis_particle_indices=(4,)).analysis_particle_indices == (4,)


def test_separate_checkpoint_file_and_opening_without_it(tmp_path):
    """tests/test_sampling.py:2130-2184: a checkpoint file NAME lives next to the analysis file; the analysis file opens for
    reading when the named checkpoint file is not there; a sampler created from a path string and one resumed from a reporter
    object see the same energies (:2186-2211)."""
    path = str(tmp_path / 'sub' / 'run.nc')
    rep = MultiStateReporter(path, checkpoint_storage='checkpoint_file.nc', open_mode='w')
    rep.close()
    assert os.path.isfile(path) and os.path.isfile(str(tmp_path / 'sub' / 'checkpoint_file.nc'))
    r = MultiStateReporter(path, checkpoint_storage='checkpoint_mod.nc', open_mode='r')       # no such checkpoint file: still opens
    assert r.read_last_iteration(last_checkpoint=False) == 0
    r.close()
    import sys
    sys.path.insert(0, HERE)
    from oracle_engine import OracleEngine
    from openmmtools_amd import testsystems, mcmc, unit
    from openmmtools_amd.multistate import ParallelTemperingSampler
    ho = testsystems.HarmonicOscillator()
    ts = states.ThermodynamicState(ho.system, 300.0 * unit.kelvin)
    ss = states.SamplerState(ho.positions, box_vectors=ho.system.getDefaultPeriodicBoxVectors())
    s = ParallelTemperingSampler(mcmc_moves=mcmc.LangevinDynamicsMove(n_steps=1), number_of_iterations=5, engine=OracleEngine(), seed=1)
    by_string = str(tmp_path / 'string.nc')
    s.create(ts, [ss], storage=by_string, min_temperature=300.0, max_temperature=400.0, n_temperatures=3)
    s.run()
    e_string = s._reporter.read_energies()[0]
    s._reporter.close()
    del s
    resumed = ParallelTemperingSampler.from_storage(MultiStateReporter(by_string), engine=OracleEngine())
    assert np.array_equal(resumed._reporter.read_energies()[0], e_string)
    assert resumed.iteration == 0                      # the last CHECKPOINT (interval 50): where the reference resumes, too


def test_write_sampler_states_like_the_references_test(tmp_path):
    """tests/test_sampling.py:636-698: checkpoints hold all particles on the interval (zero velocities when a state has none),
    the analysis file holds the flagged particles -- with their box -- every iteration, and the two agree where both exist."""
    from openmmtools_amd import testsystems
    al = testsystems.AlanineDipeptideExplicit()
    box = al.system.getDefaultPeriodicBoxVectors()
    rep = MultiStateReporter(str(tmp_path / 'w.nc'), open_mode='w', checkpoint_interval=2, analysis_particle_indices=(1, 2))
    sampler_states = [states.SamplerState(al.positions, box_vectors=box) for _ in range(2)]
    for it in range(3):
        rep.write_sampler_states(sampler_states, it)
        rep.write_last_iteration(it)
    rep.close()
    rep = MultiStateReporter(str(tmp_path / 'w.nc'), open_mode='r')
    for st, back in zip(sampler_states, rep.read_sampler_states(iteration=0)):
        assert np.allclose(st.positions, back.positions, atol=1e-6) and np.allclose(back.velocities, 0.0)
        assert np.allclose(np.asarray(box), back.box_vectors, atol=1e-6)
    analysis = rep.read_sampler_states(iteration=1, analysis_particles_only=True)
    assert type(analysis) is list and rep.read_sampler_states(iteration=1) is None
    for st in analysis:
        assert st.positions.shape == (2, 3) and st.velocities.shape == (2, 3)
    analysis, checkpoint = rep.read_sampler_states(iteration=2, analysis_particles_only=True), rep.read_sampler_states(iteration=2)
    assert len(analysis) == len(checkpoint) == 2
    for a, c in zip(analysis, checkpoint):
        assert np.allclose(a.positions, c.positions[[1, 2], :]) and np.allclose(a.velocities, c.velocities[[1, 2], :])
        assert np.allclose(a.box_vectors, c.box_vectors)


def test_store_roundtrips_of_the_references_reporter_tests(tmp_path):
    """tests/test_sampling.py:868-931, 1003-1021 on a bare reporter (netCDF4 layout, nothing else written before): state indices,
    energies with neighbourhoods and unsampled columns, mixing statistics -- written and read at once."""
    rep = MultiStateReporter(str(tmp_path / 'bare.nc'), open_mode='w')
    for i, replica_states in enumerate([[2, 1, 0, 3], np.array([3, 1, 0, 2])]):
        rep.write_replica_thermodynamic_states(replica_states, iteration=i)
        rep.write_last_iteration(i)
        assert np.all(np.asarray(replica_states) == rep.read_replica_thermodynamic_states(iteration=i))
    rep.close()
    rep = MultiStateReporter(str(tmp_path / 'bare2.nc'), open_mode='w')
    e = np.array([[0, 2, 3], [1, 2, 0], [1, 2, 3]])
    nb = np.array([[0, 1, 1], [1, 1, 0], [1, 1, 3]])
    eu = np.array([[1, 2], [2, 3.0], [3, 9.0]])
    rep.write_energies(e, nb, eu, iteration=0)
    got = rep.read_energies(iteration=0)
    assert np.all(e == got[0]) and np.all(nb == got[1]) and np.all(eu == got[2])
    acc, prop = np.array([[1, 2, 3], [4, 5, 6], [7, 8, 9]]), np.array([[3, 3, 3], [6, 6, 6], [9, 9, 9]])
    rep.write_mixing_statistics(acc, prop, iteration=0)
    back = rep.read_mixing_statistics(iteration=0)
    assert np.all(acc == back[0]) and np.all(prop == back[1])


def test_store_thermodynamic_states_one_full_serialization_per_compatible_group(tmp_path):
    """tests/test_sampling.py:513-634 (its plain-state part): states of one System are stored once, the others point at the first
    through '_Reporter__compatible_state' -- also from the unsampled group into the sampled one; all come back equal."""
    from openmmtools_amd import testsystems, unit
    from openmmtools_amd.multistate import _hdf5
    from openmmtools_amd.multistate._reference_store import _yaml_load
    lj = testsystems.LennardJonesFluid(nparticles=216).system
    other = testsystems.LennardJonesFluid(nparticles=216, epsilon=0.2 * unit.kilocalories_per_mole).system
    nvt = states.ThermodynamicState(lj, 300.0 * unit.kelvin)
    nvt_compatible = states.ThermodynamicState(lj, 320.0 * unit.kelvin)
    npt = states.ThermodynamicState(lj, 300.0 * unit.kelvin, 1.0 * unit.atmosphere)
    thermo = [nvt, nvt_compatible, npt]
    unsampled = [states.ThermodynamicState(other, 300.0 * unit.kelvin), states.ThermodynamicState(other, 300.0 * unit.kelvin),
                 states.ThermodynamicState(lj, 300.0 * unit.kelvin)]
    rep = MultiStateReporter(str(tmp_path / 'ts.nc'), open_mode='w')
    rep.write_thermodynamic_states(thermo, unsampled)
    back, back_unsampled = rep.read_thermodynamic_states()
    for a, b in zip(thermo + unsampled, back + back_unsampled):
        assert a.temperature == b.temperature and (a.pressure is None) == (b.pressure is None) and a.is_state_compatible(b)
        if a.pressure is not None:
            assert abs(a.pressure - b.pressure) < 1e-12 * a.pressure
    rep.close()
    with _hdf5.File(str(tmp_path / 'ts.nc')) as f:
        def stored(path):
            raw = f.read(path)
            text = raw.tobytes().decode() if getattr(raw, 'dtype', None) is not None and raw.dtype.kind == 'S' else str(np.asarray(raw).reshape(-1)[0])
            return _yaml_load(text)
        s = [stored('/thermodynamic_states/state%d' % k) for k in range(3)]
        u = [stored('/unsampled_states/state%d' % k) for k in range(3)]
    assert 'standard_system' in s[0] and 'standard_system' not in s[1] and s[1]['_Reporter__compatible_state'] == 'thermodynamic_states/0'
    assert 'standard_system' in s[2]                                          # NPT: another ensemble, its own serialization
    assert 'standard_system' in u[0] and u[1]['_Reporter__compatible_state'] == 'unsampled_states/0'
    assert u[2]['_Reporter__compatible_state'] == 'thermodynamic_states/0'


def test_small_accessors_of_the_references_reporter(tmp_path):
    """multistatereporter.py:197-215 n_states / n_replicas / is_periodic, :480-560 read_end_thermodynamic_states, :1203-1234
    read_logZ / write_logZ: on the store the reference wrote, on a store of the reference's layout written here, and on the record
    container."""
    import sys
    sys.path.insert(0, HERE)
    from oracle_engine import OracleEngine
    from openmmtools_amd import testsystems, mcmc, unit
    from openmmtools_amd.multistate import SAMSSampler
    rep = MultiStateReporter(STORE, open_mode='r')
    assert (rep.n_states, rep.n_replicas, rep.is_periodic) == (20, 1, True)
    ends = rep.read_end_thermodynamic_states()                              # no unsampled states: first and last sampled one
    assert len(ends) == 2 and ends[0].temperature == 300.0 and abs(ends[1].temperature - 600.0) < 1e-9
    rep.close()
    assert rep.n_states is None and rep.n_replicas is None and rep.is_periodic is None
    ho = testsystems.HarmonicOscillator()
    sts = [states.ThermodynamicState(ho.system, T * unit.kelvin) for T in np.linspace(300.0, 500.0, 5)]
    unsampled = [states.ThermodynamicState(ho.system, 700.0 * unit.kelvin)]
    ss = states.SamplerState(ho.positions)
    move = mcmc.LangevinSplittingDynamicsMove(timestep=2.0 * unit.femtosecond, collision_rate=5.0 / unit.picosecond, n_steps=5,
                                              reassign_velocities=True, splitting='V R O R V')
    for name in ('layout.nc', 'records'):
        s = SAMSSampler(mcmc_moves=move, number_of_iterations=3, engine=OracleEngine(), seed=3)
        s.create(sts, [ss] * 2, storage=MultiStateReporter(str(tmp_path / name), checkpoint_interval=1), unsampled_thermodynamic_states=unsampled)
        s.run()
        s._reporter.close()
        r = MultiStateReporter(str(tmp_path / name), open_mode='r')
        periodic = s.sampler_states[0].box_vectors is not None                 # (what the sampler holds is what is stored)
        assert (r.n_states, r.n_replicas, r.is_periodic) == (5, 2, periodic), name
        ends = r.read_end_thermodynamic_states()
        assert len(ends) == 1 and abs(ends[0].temperature - 700.0) < 1e-9, name
        assert np.array_equal(np.asarray(r.read_logZ(3)), np.asarray(s._logZ)), name
        r.close()
        w = MultiStateReporter(str(tmp_path / name), open_mode='a')
        w.write_logZ(3, np.arange(5.0))
        assert np.array_equal(np.asarray(w.read_logZ(3)), np.arange(5.0)), name
        w.close()

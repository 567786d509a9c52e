"""Storage write path and resume (openmmtools/multistate/multistatereporter.py, from_storage
multistatesampler.py:263-299).  The reference's own storage tests check that what is read back equals what the sampler
held (tests/test_sampling.py:1139-1231 stored energies / states, :1380-1460 resume)."""
import numpy as np
import pytest
from openmmtools_amd import testsystems, states, mcmc, unit
from openmmtools_amd.multistate import (ParallelTemperingSampler, ReplicaExchangeSampler, SAMSSampler,
                                        MultiStateReporter, MultiStateSampler)
from oracle_engine import OracleEngine


def _move(n=10):
    return mcmc.LangevinSplittingDynamicsMove(timestep=1.0 * unit.femtosecond, collision_rate=1.0 / unit.picosecond,
                                              n_steps=n, reassign_velocities=True, splitting='V R O R V')


def _pt(tmp_path, n_iter, interval=2, storage=True):
    ho = testsystems.HarmonicOscillator()
    ts = states.ThermodynamicState(ho.system, 300.0)
    ss = states.SamplerState(ho.positions)
    rep = MultiStateReporter(str(tmp_path / 'pt_store'), checkpoint_interval=interval) if storage else None
    s = ParallelTemperingSampler(mcmc_moves=_move(), number_of_iterations=n_iter, engine=OracleEngine(), seed=11)
    s.create(ts, [ss], storage=rep, min_temperature=300.0, max_temperature=600.0, n_temperatures=4)
    return s, rep


def test_every_iteration_is_stored_with_reference_dtypes(tmp_path):
    s, rep = _pt(tmp_path, 5)
    seen = []
    for _ in range(5):
        s.run(1)
        seen.append((s.energy_thermodynamic_states.copy(), s.replica_thermodynamic_states.copy(),
                     s._n_accepted_matrix.copy(), s._n_proposed_matrix.copy()))
    r = MultiStateReporter(str(tmp_path / 'pt_store'), open_mode='r')
    e, nb, eu = r.read_energies()
    assert e.shape == (6, 4, 4) and e.dtype == np.dtype('<f8') and nb.dtype == np.dtype('i1') and eu.shape == (6, 4, 0)
    st = r.read_replica_thermodynamic_states()
    acc, prop = r.read_mixing_statistics()
    assert acc.dtype == np.dtype('<i4') and acc.shape == (6, 4, 4)
    for it, (ue, lab, a, p) in enumerate(seen, start=1):
        assert np.array_equal(e[it], ue) and np.array_equal(st[it], lab)
        assert np.array_equal(acc[it], a) and np.array_equal(prop[it], p)
    assert np.all(nb == 1) and np.isfinite(e[0]).all()            # iteration 0 = initial energies
    assert r.read_last_iteration(last_checkpoint=False) == 5
    assert r.read_checkpoint_iterations() == [0, 2, 4] and r.read_last_iteration() == 4
    assert r.read_sampler_states(3) is None                        # not a checkpoint iteration
    cp = r.read_sampler_states(4)
    assert len(cp) == 4 and cp[0].positions.shape == (1, 3)
    assert np.all(np.diff(r.read_timestamp()) >= 0)


def test_checkpoint_positions_are_float32_of_the_sampler_state(tmp_path):
    s, rep = _pt(tmp_path, 2, interval=1)
    s.run()
    x = np.stack([st.positions for st in s.sampler_states])
    cp = MultiStateReporter(str(tmp_path / 'pt_store'), open_mode='r').read_sampler_states(2)
    xs = np.stack([c.positions for c in cp])
    assert np.array_equal(xs, x.astype(np.float32).astype(np.float64))
    assert not np.array_equal(xs, x)                               # f4 on disk (multistatereporter.py:1621-1632)


def test_resume_from_storage_continues_the_same_markov_chain(tmp_path):
    """A run interrupted after a checkpoint and resumed with from_storage gives exactly the stored-precision
    continuation: labels, statistics and energies of the later iterations equal those of a sampler restarted from
    the same f4 checkpoint by hand."""
    s, rep = _pt(tmp_path, 6, interval=2)
    s.run(4)                                                       # iterations 1..4, checkpoint at 4
    del s
    r = ParallelTemperingSampler.from_storage(str(tmp_path / 'pt_store'), engine=OracleEngine())
    assert r.iteration == 4 and r.number_of_iterations == 6 and type(r) is ParallelTemperingSampler
    assert np.array_equal(r.replica_thermodynamic_states,
                          MultiStateReporter(str(tmp_path / 'pt_store'), open_mode='r').read_replica_thermodynamic_states(4))
    r.run()
    assert r.iteration == 6 and r.is_completed
    rd = MultiStateReporter(str(tmp_path / 'pt_store'), open_mode='r')
    e, _, _ = rd.read_energies()
    assert e.shape[0] == 7 and rd.read_last_iteration(last_checkpoint=False) == 6
    # the resumed chain is a deterministic function of the checkpoint: a second resume reproduces it bit for bit
    import shutil
    shutil.copytree(str(tmp_path / 'pt_store'), str(tmp_path / 'copy_store'))
    shutil.copytree(str(tmp_path / 'pt_store_checkpoint'), str(tmp_path / 'copy_store_checkpoint'))
    rc = MultiStateReporter(str(tmp_path / 'copy_store'), open_mode='a')
    rc.write_last_iteration(4)
    r2 = ParallelTemperingSampler.from_storage(rc, engine=OracleEngine())
    r2.run()
    e2, _, _ = MultiStateReporter(str(tmp_path / 'copy_store'), open_mode='r').read_energies()
    assert np.array_equal(e2[5:], e[5:])
    with pytest.raises(TypeError):
        SAMSSampler.from_storage(str(tmp_path / 'pt_store'), engine=OracleEngine())


def test_sams_online_data_and_resume(tmp_path):
    ho = testsystems.HarmonicOscillator()
    tss = [states.ThermodynamicState(ho.system, T) for T in (300.0, 350.0, 400.0)]
    ss = states.SamplerState(ho.positions)
    rep = MultiStateReporter(str(tmp_path / 'sams'), checkpoint_interval=3)
    s = SAMSSampler(mcmc_moves=_move(5), number_of_iterations=9, engine=OracleEngine(), seed=2)
    s.create(tss, [ss], storage=rep)
    s.run(6)
    logZ6, hist6 = s._logZ.copy(), s._state_histogram.copy()
    rd = MultiStateReporter(str(tmp_path / 'sams'), open_mode='r')
    online = rd.read_online_data_if_present(6)
    assert np.array_equal(online['logZ'], logZ6) and np.array_equal(online['sams_state']['histogram'], hist6)
    r = MultiStateSampler.from_storage(str(tmp_path / 'sams'), engine=OracleEngine())
    assert type(r) is SAMSSampler and r.iteration == 6
    assert np.array_equal(r._logZ, logZ6) and np.array_equal(r._state_histogram, hist6)
    r.run()
    assert r.iteration == 9
    # sams.py:381-393 / tests/test_sampling.py:2757-2787: the histogram equals np.histogram of the stored labels
    lab = MultiStateReporter(str(tmp_path / 'sams'), open_mode='r').read_replica_thermodynamic_states()
    counts = np.bincount(lab[:, 0], minlength=3)              # every stored iteration, the initial one included (reference)
    assert np.array_equal(r._state_histogram, counts)


def test_storage_path_string_and_write_mode_guard(tmp_path):
    ho = testsystems.HarmonicOscillator()
    ts = [states.ThermodynamicState(ho.system, T) for T in (300.0, 400.0)]
    s = ReplicaExchangeSampler(mcmc_moves=_move(3), number_of_iterations=2, engine=OracleEngine(), seed=5)
    s.create(ts, [states.SamplerState(ho.positions)], storage=str(tmp_path / 'rex'))
    s.run()
    r = MultiStateReporter(str(tmp_path / 'rex'), open_mode='r')
    assert r.read_energies()[0].shape == (3, 2, 2)
    with pytest.raises(IOError):
        r.write_last_iteration(1)
    with pytest.raises(IOError):
        MultiStateReporter(str(tmp_path / 'missing'), open_mode='r')


def test_create_equilibrate_run_resume_keeps_iteration_zero(tmp_path):
    """multistatesampler.py:588-609, 738-753: iteration 0 is reported at create() and its energies are rewritten by the
    first run() whatever happened in between; a resume before the next checkpoint must find the initial permutation."""
    s, rep = _pt(tmp_path, 4, interval=10)
    r0 = MultiStateReporter(str(tmp_path / 'pt_store'), open_mode='r')
    assert r0.read_last_iteration(last_checkpoint=False) == 0            # on disk right after create()
    assert r0.read_replica_thermodynamic_states(0).tolist() == [0, 1, 2, 3]
    s.equilibrate(2)
    labels_after_equil = s.replica_thermodynamic_states.copy()
    s.run(3)                                                             # no checkpoint after iteration 0 (interval 10)
    rd = MultiStateReporter(str(tmp_path / 'pt_store'), open_mode='r')
    e, nb, _ = rd.read_energies()
    assert np.all(nb[0] == 1) and np.isfinite(e[0]).all() and np.abs(e[0]).sum() > 0
    assert np.array_equal(rd.read_replica_thermodynamic_states(0), labels_after_equil)
    x0 = np.stack([c.positions for c in rd.read_sampler_states(0)])
    assert np.abs(x0).sum() > 0                                          # the equilibrated positions, not the initial zeros
    del s
    r = ParallelTemperingSampler.from_storage(str(tmp_path / 'pt_store'), engine=OracleEngine())
    assert r.iteration == 0
    assert np.array_equal(r.replica_thermodynamic_states, labels_after_equil)
    assert sorted(r.replica_thermodynamic_states.tolist()) == [0, 1, 2, 3]
    r.run(2)
    assert r.iteration == 2 and sorted(r.replica_thermodynamic_states.tolist()) == [0, 1, 2, 3]


def test_create_refuses_to_overwrite_and_open_w_spares_foreign_files(tmp_path):
    s, rep = _pt(tmp_path, 2)
    s.run(1)
    with pytest.raises(RuntimeError, match='refusing to overwrite'):
        _pt(tmp_path, 2)                                                 # multistatesampler.py:588
    d = tmp_path / 'other'
    d.mkdir()
    (d / 'notes.txt').write_text('mine')
    (d / 'energies.f8').write_bytes(b'')
    # a user's own *.json / *.yaml / *.npz / *.pkl in a directory without meta.json are not this format's files
    for name in ('config.json', 'protocol.yaml', 'frames.npz', 'model.pkl', 'energies.f8.bak'):
        (d / name).write_text('keep')
    MultiStateReporter(str(d), open_mode='w')
    assert (d / 'notes.txt').read_text() == 'mine' and not (d / 'energies.f8').exists()
    for name in ('config.json', 'protocol.yaml', 'frames.npz', 'model.pkl', 'energies.f8.bak'):
        assert (d / name).read_text() == 'keep'


def test_storage_objects_are_unpickled_with_a_whitelist(tmp_path):
    import pickle, subprocess
    s, rep = _pt(tmp_path, 2)
    s.run(1)
    with open(str(tmp_path / 'pt_store' / 'metadata.pkl'), 'wb') as fh:
        pickle.dump(subprocess.Popen, fh)                               # a class reference the format never stores
    with pytest.raises(pickle.UnpicklingError):
        MultiStateReporter(str(tmp_path / 'pt_store'), open_mode='r').read_dict('metadata')
    # from protocol 4 on STACK_GLOBAL hands find_class a dotted name that getattr walks ('os.system' through any
    # module of this package that imports os); and a whole-numpy whitelist admits exec gadgets
    marker = tmp_path / 'pwned'
    def stack_global(module, name, arg=None):
        """protocol-4 pickle: push module and (possibly dotted) name, STACK_GLOBAL, optionally call with one string argument"""
        def u(t):
            return b'\x8c' + bytes([len(t)]) + t.encode()
        out = b'\x80\x04' + u(module) + u(name) + b'\x93'
        if arg is not None:
            out += u(arg) + b'\x85R'
        return out + b'.'
    payloads = [
        stack_global('openmmtools_amd.multistate.multistatereporter', 'os.system', 'touch %s' % marker),
        stack_global('openmmtools_amd.multistate.multistatereporter', 'os'),
        pickle.dumps(subprocess.check_call),
        stack_global('numpy.testing._private.utils', 'runstring'),
        stack_global('numpy', 'load'),
        stack_global('builtins', 'eval', '1'),
    ]
    for payload in payloads:
        with open(str(tmp_path / 'pt_store' / 'metadata.pkl'), 'wb') as fh:
            fh.write(payload)
        with pytest.raises(pickle.UnpicklingError):
            MultiStateReporter(str(tmp_path / 'pt_store'), open_mode='r').read_dict('metadata')
    assert not marker.exists()
    # what the format does store still loads: this package's classes, numpy arrays and scalars, builtin containers
    import numpy as np
    rep2 = MultiStateReporter(str(tmp_path / 'pt_store'), open_mode='a')
    rep2.write_dict('metadata', dict(a=np.arange(3.0), b=np.float64(2.5), c=[1, (2, 3)], d={'x': None}))
    back = rep2.read_dict('metadata')
    assert np.array_equal(back['a'], np.arange(3.0)) and back['b'] == 2.5 and back['c'] == [1, (2, 3)]
    assert len(rep2.read_thermodynamic_states()[0]) == 4 and len(rep2.read_mcmc_moves()) == 4


class _UserSampler(ParallelTemperingSampler):
    """a sampler class defined outside the package, as a user's script would"""


def test_a_users_subclass_resumes_through_its_own_from_storage(tmp_path):
    """from_storage refused every sampler class whose module is not under openmmtools_amd.  A subclass resumes through
    ITS OWN from_storage (the caller supplies the class; nothing named in the storage is imported); any other foreign name is
    still refused."""
    ho = testsystems.HarmonicOscillator()
    ts_ = states.ThermodynamicState(ho.system, 300.0)
    ss = states.SamplerState(ho.positions)
    rep = MultiStateReporter(str(tmp_path / 'u_store'), checkpoint_interval=1)
    s = _UserSampler(mcmc_moves=_move(), number_of_iterations=4, engine=OracleEngine(), seed=11)
    s.create(ts_, [ss], storage=rep, min_temperature=300.0, max_temperature=600.0, n_temperatures=3)
    s.run(2)
    r = _UserSampler.from_storage(str(tmp_path / 'u_store'), engine=OracleEngine())
    assert type(r) is _UserSampler and r.iteration == 2
    r.run(1)
    with pytest.raises(TypeError, match='resume with that class'):
        ParallelTemperingSampler.from_storage(str(tmp_path / 'u_store'), engine=OracleEngine())


@pytest.mark.parametrize('name', ['store', 'store.nc'])
def test_last_iteration_functions(tmp_path, name):
    """tests/test_sampling.py:2080-2128 on both layouts: after the last good iteration is set back to 4 of 10, reads by index,
    negative index, slice and reversed slice are relative to iteration 4, and an index beyond it raises IndexError."""
    import sys, os
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from oracle_engine import OracleEngine
    from openmmtools_amd import testsystems, states, mcmc, unit
    from openmmtools_amd.multistate import ParallelTemperingSampler, MultiStateReporter
    ho = testsystems.HarmonicOscillator()
    ts = states.ThermodynamicState(ho.system, 300.0 * unit.kelvin)
    ss = states.SamplerState(ho.positions, box_vectors=ho.system.getDefaultPeriodicBoxVectors())
    move = mcmc.LangevinDynamicsMove(timestep=1.0 * unit.femtosecond, n_steps=1)
    s = ParallelTemperingSampler(mcmc_moves=move, number_of_iterations=10, engine=OracleEngine(), seed=2, online_analysis_interval=None)
    rep = MultiStateReporter(str(tmp_path / name), checkpoint_interval=2)
    s.create(ts, [ss], storage=rep, min_temperature=300.0, max_temperature=400.0, n_temperatures=3)
    s.run()
    rep.close()
    rep = MultiStateReporter(str(tmp_path / name), open_mode='a', checkpoint_interval=2)
    all_energies = rep.read_energies()[0]
    all_states = rep.read_replica_thermodynamic_states()
    assert all_energies.shape[0] == 11
    rep.write_last_iteration(4)                                  # "break the checkpoint" (:2108-2110)
    rep.close()
    rep = MultiStateReporter(str(tmp_path / name), open_mode='r', checkpoint_interval=2)
    assert np.all(rep.read_energies(1)[0] == all_energies[1])
    assert np.all(rep.read_energies(-1)[0] == all_energies[4])
    assert np.all(rep.read_energies()[0] == all_energies[:5])
    assert np.all(rep.read_energies(slice(-1, None, -1))[0] == all_energies[4::-1])
    assert np.all(rep.read_replica_thermodynamic_states(-1) == all_states[4]) and rep.read_mixing_statistics()[0].shape[0] == 5
    with pytest.raises(IndexError):
        rep.read_energies(7)


@pytest.mark.parametrize('storage', ['dicts', 'dicts.nc'])
def test_store_dict(tmp_path, storage):
    """tests/test_sampling.py:934-1001 (multistatereporter.py:1094-1165, 1817-1880) on both layouts: booleans, strings, numbers,
    lists, (nested) numpy arrays and nested dictionaries come back equal from the single-string, the nested and the
    fixed-dimension representation; an entry can be read by its path; a rewrite with the same structure updates in place."""
    from openmmtools_amd.multistate import MultiStateReporter
    data = {'mybool': False, 'mystring': 'test', 'myinteger': 3, 'myfloat': 4.0, 'mylist': [0, 1, 2, 3],
            'mynumpyarray': np.array([2.0, 3, 4]), 'mynestednumpyarray': np.array([[1, 2, 3], [4.0, 5, 6]]),
            'mynesteddict': {'field1': 'string', 'field2': {'field21': 3.0, 'field22': True}}}

    def same(a, b):
        if isinstance(a, dict):
            return isinstance(b, dict) and sorted(a) == sorted(b) and all(same(a[k], b[k]) for k in a)
        if isinstance(a, np.ndarray):
            return isinstance(b, np.ndarray) and a.dtype == b.dtype and np.array_equal(a, b)
        return type(a) is type(b) and a == b
    rep = MultiStateReporter(str(tmp_path / storage), open_mode='w')
    if not storage.endswith('.nc'):
        rep.initialize(1, 1, 0, 1)
    for name, kwargs in (('testdict', {}), ('nested', dict(nested=True)), ('fixed', dict(fixed_dimension=True))):
        rep._write_dict(name, data, **kwargs)
        assert same(data, rep.read_dict(name)), name
        assert same(data['mynesteddict']['field2'], rep.read_dict(name + '/mynesteddict/field2'))
        if name != 'fixed':                                   # (a fixed-length text cannot change its length)
            changed = dict(data, mybool=True, mystring='substituted')
            rep._write_dict(name, changed, **kwargs)
            back = rep.read_dict(name)
            assert back['mybool'] is True and back['mystring'] == 'substituted'
    rep.close()
    if storage.endswith('.nc'):
        from openmmtools_amd.multistate import _hdf5
        with _hdf5.File(str(tmp_path / storage)) as f:         # :990-996: groups and variables of the nested form, one variable otherwise
            assert f.is_group('/nested') and f.is_group('/nested/mynesteddict') and '/nested/mylist' in f
            assert '/testdict' in f and not f.is_group('/testdict')


@pytest.mark.parametrize('storage', ['props', 'props.nc'])
def test_stored_properties_stay_in_sync_with_the_storage(tmp_path, storage):
    """tests/test_sampling.py:1548-1605: options assigned after create() reach the stored 'options', new sampler states assigned
    before run() reach the checkpoint of iteration 0."""
    import sys, os
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from oracle_engine import OracleEngine
    from openmmtools_amd import testsystems, states, mcmc, unit
    from openmmtools_amd.multistate import ParallelTemperingSampler, MultiStateReporter
    ho = testsystems.HarmonicOscillator()
    ts = states.ThermodynamicState(ho.system, 300.0 * unit.kelvin)
    ss = states.SamplerState(ho.positions, box_vectors=ho.system.getDefaultPeriodicBoxVectors())
    move = mcmc.LangevinDynamicsMove(timestep=1.0 * unit.femtosecond, n_steps=1)
    s = ParallelTemperingSampler(mcmc_moves=move, number_of_iterations=5, online_analysis_interval=1, engine=OracleEngine(), seed=2)
    rep = MultiStateReporter(str(tmp_path / storage), checkpoint_interval=1)
    s.create(ts, [ss], storage=rep, min_temperature=300.0, max_temperature=400.0, n_temperatures=3)
    s.number_of_iterations = float('inf')
    s.online_analysis_interval = 7
    moved = s.sampler_states
    original = moved[0].positions.copy()
    moved[0].positions = original + 0.1
    s.sampler_states = moved
    rep.close()
    back = MultiStateReporter(str(tmp_path / storage), open_mode='r')
    opts = back.read_dict('options')
    flat = opts if 'kwargs' not in opts else dict(opts['kwargs'], number_of_iterations=opts['number_of_iterations'])
    assert flat['number_of_iterations'] == float('inf') and flat['online_analysis_interval'] == 7
    assert np.allclose(back.read_sampler_states(0)[0].positions, original + 0.1, atol=1e-6)


@pytest.mark.parametrize('storage', ['ckpt', 'ckpt.nc'])
def test_checkpointing_writes_on_the_interval_only(tmp_path, storage):
    """tests/test_sampling.py:1997-2030: energies exist for every iteration, sampler states only on the checkpoint interval."""
    import sys, os
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from oracle_engine import OracleEngine
    from openmmtools_amd import testsystems, states, mcmc, unit, integrators
    from openmmtools_amd.multistate import ParallelTemperingSampler, MultiStateReporter
    ho = testsystems.HarmonicOscillator()
    ts = states.ThermodynamicState(ho.system, 300.0 * unit.kelvin)
    ss = states.SamplerState(ho.positions, box_vectors=ho.system.getDefaultPeriodicBoxVectors())
    move = mcmc.IntegratorMove(integrators.VelocityVerletIntegrator(1.0 * unit.femtosecond), n_steps=1)    # the reference's VerletIntegrator move
    rep = MultiStateReporter(str(tmp_path / storage), checkpoint_interval=2)
    s = ParallelTemperingSampler(mcmc_moves=move, number_of_iterations=3, engine=OracleEngine(), seed=2, online_analysis_interval=None)
    s.create(ts, [ss], storage=rep, min_temperature=300.0, max_temperature=400.0, n_temperatures=3)
    s.run()
    rep.close()
    rep = MultiStateReporter(str(tmp_path / storage), open_mode='r', checkpoint_interval=2)
    for i in range(3):
        energies = rep.read_energies(i)[0]
        got = rep.read_sampler_states(i)
        assert type(energies) is np.ndarray and energies.shape == (3, 3)
        if rep._calculate_checkpoint_iteration(i) is not None:
            assert got[0].positions.shape == (1, 3)
        else:
            assert got is None
    assert [rep._calculate_checkpoint_iteration(i) for i in range(5)] == [0, None, 1, None, 2]

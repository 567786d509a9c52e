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
    # "delete reporters and load again": the continuation is a store of this package's format holding the whole history
    del sampler
    rep = MultiStateReporter(str(tmp_path / 'continued.nc'), open_mode='r')
    assert not rep.is_reference_store and rep.read_last_iteration() == 3
    e, nb, eu = rep.read_energies()
    ref = MultiStateReporter(STORE, open_mode='r')
    assert e.shape == (4, 1, 20) and np.array_equal(e[:3], ref.read_energies()[0])
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

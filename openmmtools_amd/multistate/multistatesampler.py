"""MultiStateSampler: the replica-exchange iteration loop driving the device engine.

Mirrors openmmtools/multistate/multistatesampler.py (class :63): ``create`` (:537-609,
_pre_write_create :836-926), ``run`` (:724-804: mix -> propagate -> energies), ``equilibrate``
(:649-722: propagate -> energies -> mix), and the three hooks the engine replaces:
``_mix_replicas`` (:1500-1517), ``_propagate_replicas`` (:1287-1337), ``_compute_energies``
(:1436-1494).  ``storage`` (a path or a MultiStateReporter) receives every iteration's energies, state
indices and mixing statistics plus f4 checkpoints (multistatereporter.py); ``from_storage`` resumes from the
last complete checkpoint.  Online analysis and minimization are out of scope.
"""
import collections
import copy
import os
import time
import logging
import numpy as np

from .. import mcmc, unit
from ..utils import with_timer
from ..system import system_to_desc
from .utils import SimulationNaNError
from .comm import SingleProcessComm

logger = logging.getLogger(__name__)


class MultiStateSampler:
    def __init__(self, mcmc_moves=None, number_of_iterations=1, locality=None,
                 online_analysis_interval=200, online_analysis_target_error=0.0,
                 online_analysis_minimum_iterations=200, engine=None, seed=0xC0FFEE, comm=None):
        if locality is not None and ((type(locality) != int) or (locality <= 0)):      # :504-508
            raise ValueError('locality must be an int > 0')
        # multistatesampler.py:478-501
        if online_analysis_interval is not None and (type(online_analysis_interval) != int or online_analysis_interval < 1):
            raise ValueError('online_analysis_interval must be an integer >=1 or None')
        if online_analysis_interval is not None:
            if online_analysis_target_error < 0:
                raise ValueError('online_analysis_target_error must be a float >= 0')
            if type(online_analysis_minimum_iterations) is not int or online_analysis_minimum_iterations < 0:
                raise ValueError('online_analysis_minimum_iterations must be an integer >= 0')
        if not (0 <= number_of_iterations <= float('inf')):                              # :469-475 (the text is the reference's, run-on included)
            raise ValueError('Accepted values for number_of_iterations are' 'non-negative integers and infinity.')
        if mcmc_moves is None:
            # multistatesampler.py:224-227
            self._mcmc_moves = mcmc.LangevinDynamicsMove(timestep=2.0 * unit.femtosecond,
                                                         collision_rate=5.0 / unit.picosecond,
                                                         n_steps=500, reassign_velocities=True,
                                                         n_restart_attempts=6)
        else:
            self._mcmc_moves = copy.deepcopy(mcmc_moves)
        self.number_of_iterations = number_of_iterations
        self.locality = locality
        self.online_analysis_interval = online_analysis_interval
        self.online_analysis_target_error = online_analysis_target_error
        self.online_analysis_minimum_iterations = online_analysis_minimum_iterations
        self._last_mbar_f_k = None                # :246-247
        self._last_err_free_energy = None
        self._engine = engine
        self._seed = int(seed)
        self._comm = comm if comm is not None else SingleProcessComm()
        self._thermodynamic_states = None
        self._unsampled_states = None
        self._sampler_states = None
        self._replica_thermodynamic_states = None
        self._iteration = None
        self._energy_thermodynamic_states = None
        self._neighborhoods = None
        self._energy_unsampled_states = None
        self._n_accepted_matrix = None
        self._n_proposed_matrix = None
        self._reporter = None
        self._timing_data = dict()
        self._sampler_states_stale = False
        self._device_ukl = None          # torch tensor [R, K_total] (multi-rank or device mixing)
        self.verify_labels = False

    # ---- properties (multistatesampler.py:301-436) ----------------------------------------
    @property
    def n_states(self):
        return 0 if self._thermodynamic_states is None else len(self._thermodynamic_states)

    @property
    def n_replicas(self):
        return 0 if self._sampler_states is None else len(self._sampler_states)

    @property
    def iteration(self):
        return self._iteration

    @property
    def mcmc_moves(self):
        return copy.deepcopy(self._mcmc_moves)

    @mcmc_moves.setter
    def mcmc_moves(self, new_value):
        """:398-408: only before create() (a single move becomes one per state there)."""
        if self._thermodynamic_states is not None:
            raise RuntimeError('Cannot modify MCMCMoves after creation.')
        self._mcmc_moves = copy.deepcopy(new_value)

    @property
    def thermodynamic_states(self):
        return self._thermodynamic_states

    @property
    def sampler_states(self):
        self._sync_sampler_states()
        return self._sampler_states

    @sampler_states.setter
    def sampler_states(self, value):
        """:417-429: new configurations between create() and run(); they go to the engine and to the storage."""
        if self._iteration != 0:
            raise RuntimeError('Sampler states can be assigned only between create() and run().')
        if len(value) != self.n_replicas:
            raise ValueError('Passed {} sampler states for {} replicas'.format(len(value), self.n_replicas))
        self._sampler_states = copy.deepcopy(list(value))
        self._sampler_states_stale = False
        sl = slice(self._r_begin, self._r_begin + self._r_count)
        x = np.stack([s.positions for s in self._sampler_states[sl]])
        have_v = all(s.velocities is not None for s in self._sampler_states[sl])
        v = np.stack([s.velocities for s in self._sampler_states[sl]]) if have_v else None
        if self._thermodynamic_states[0].is_periodic:
            box = np.stack([s.box_edges for s in self._sampler_states[sl]])
        else:
            box = np.zeros((self._r_count, 3))
        self._engine.set_replicas(self.n_replicas, self._r_begin, x, v, box, self._replica_thermodynamic_states)
        self._compute_energies()                       # the stored energies of iteration 0 belong to the configurations in place
        if self._reporter is not None and self._comm.rank == 0:
            self._reporter.write_sampler_states(self._sampler_states, self._iteration)
            self._reporter.write_energies(self._energy_thermodynamic_states, self._neighborhoods, self._energy_unsampled_states, self._iteration)

    @classmethod
    def default_options(cls):
        """:1224-1237: the keyword defaults of __init__ along the class hierarchy (without ``mcmc_moves``)."""
        import inspect
        out = {}
        for c in inspect.getmro(cls):
            if c is object:
                continue
            spec = inspect.getfullargspec(c.__init__)
            if spec.defaults:
                out.update(dict(zip(spec.args[-len(spec.defaults):], spec.defaults)))
        out.pop('mcmc_moves', None)
        for private in ('engine', 'comm', 'seed'):      # this package's own constructor arguments, not simulation options
            out.pop(private, None)
        return out

    # multistatesampler.py:129-131, 1755-1764: the reference propagates and evaluates energies in two ContextCaches; here one
    # engine handle per GPU plays both parts (``engine=``).  The attributes exist so that scripts which assign caches keep
    # running; what is assigned is kept and not used.
    @property
    def energy_context_cache(self):
        from .. import cache
        own = self.__dict__.get('_energy_context_cache')             # (a cache is falsy while empty: compare with None)
        return cache.global_context_cache if own is None else own     # :1763-1764: the global cache by default

    @energy_context_cache.setter
    def energy_context_cache(self, value):
        self.__dict__['_energy_context_cache'] = value

    @property
    def sampler_context_cache(self):
        from .. import cache
        own = self.__dict__.get('_sampler_context_cache')
        return cache.global_context_cache if own is None else own

    @sampler_context_cache.setter
    def sampler_context_cache(self, value):
        self.__dict__['_sampler_context_cache'] = value

    @property
    def replica_thermodynamic_states(self):
        return self._replica_thermodynamic_states

    @property
    def energy_thermodynamic_states(self):
        return self._energy_thermodynamic_states

    @property
    def is_periodic(self):
        """multistatesampler.py:431-436."""
        if self._sampler_states is None:
            return None
        return self._thermodynamic_states[0].is_periodic

    @property
    def metadata(self):
        """:519-523."""
        return copy.deepcopy(getattr(self, '_metadata', None))

    @property
    def options(self):
        """What is stored for ``from_storage`` / ``read_status`` (:1145-1167)."""
        o = self._options()
        flat = dict(number_of_iterations=o['number_of_iterations'])
        flat.update(o['kwargs'])
        return flat

    @property
    def engine(self):
        return self._engine

    def __repr__(self):
        return '<instance of {}>'.format(self.__class__.__name__)

    class Status(collections.namedtuple('Status', ['iteration', 'target_error', 'is_completed'])):
        """:301-305."""

    @classmethod
    def read_status(cls, storage):
        """:307-358: (iteration, target_error, is_completed) from the storage alone, without building the sampler."""
        from .multistatereporter import MultiStateReporter
        rep = MultiStateReporter(storage) if isinstance(storage, (str, bytes, os.PathLike)) else storage
        was_open = rep.is_open()
        if not was_open:
            if not rep.storage_exists():                                        # :1163-1166
                raise FileNotFoundError('Storage file {} or its subfiles do not exist; cannot read status.'.format(rep.filepath))
            rep.open('r')
        try:
            opts = rep.read_dict('options')
            kw = opts['kwargs']
            iteration = rep.read_last_iteration(last_checkpoint=False) or 0     # nothing reported yet: iteration 0
            target_error = last_err = None
            if kw.get('online_analysis_interval') is not None and kw.get('online_analysis_target_error', 0.0) != 0.0:
                target_error = kw['online_analysis_target_error']
                try:
                    last_err = float(cls._read_last_free_energy(rep, iteration)[1][1])
                except TypeError:
                    last_err = np.inf                                      # no free energy stored yet
        finally:
            if not was_open:
                rep.close()
        done = cls._is_completed_static(opts['number_of_iterations'], iteration, last_err, kw.get('online_analysis_target_error', 0.0))
        return cls.Status(iteration=iteration, target_error=target_error, is_completed=done)

    def extend(self, n_iterations):
        """:806-822: like run(), but raises ``number_of_iterations`` when needed (and stores the new value)."""
        if self._iteration + n_iterations > self.number_of_iterations:
            self.number_of_iterations = self._iteration + n_iterations
            if self._reporter is not None and self._comm.rank == 0:
                self._reporter.write_dict('options', self._options())
        self.run(n_iterations)

    # ---- create ---------------------------------------------------------------------------
    @classmethod
    def _default_initial_thermodynamic_states(cls, thermodynamic_states, sampler_states):
        """multistatesampler.py:1117-1143."""
        n_thermo, n_sampler = len(thermodynamic_states), len(sampler_states)
        thermo_indices = np.arange(n_thermo, dtype=int)
        initial = np.zeros(n_sampler, dtype=int)
        loops = n_sampler // n_thermo
        n_looped = n_thermo * loops
        initial[:n_looped] = np.tile(thermo_indices, loops)
        initial[n_looped:] = np.linspace(0, n_thermo - 1, n_sampler - n_looped, dtype=int)
        return initial

    def create(self, thermodynamic_states, sampler_states, storage=None, initial_thermodynamic_states=None,
               unsampled_thermodynamic_states=None, metadata=None):
        if self._thermodynamic_states is not None:
            raise RuntimeError('Cannot invoke create() on an already initialized sampler')   # :572-574
        if hasattr(sampler_states, 'positions'):
            sampler_states = [sampler_states]
        self._pre_write_create(thermodynamic_states, sampler_states, storage,
                               initial_thermodynamic_states=initial_thermodynamic_states,
                               unsampled_thermodynamic_states=unsampled_thermodynamic_states, metadata=metadata)
        self._initialize_reporter(storage)
        self._initialize_engine()
        self._iteration0_energies_reported = False
        # multistatesampler.py:588-609 (_initialize_reporter -> _report_iteration): iteration 0 -- initial positions, state
        # indices, zeroed energies -- is on disk from creation on (SAMS counts the initial states in its histogram here,
        # sams.py:381-393); run() rewrites the energies of iteration 0 (:738-753)
        self._report_iteration()

    def _initialize_reporter(self, storage):
        """multistatesampler.py:1169-1187: a path or a MultiStateReporter; states, moves, options and metadata are
        stored at creation, iteration 0 (initial energies) by the first run()."""
        from .multistatereporter import MultiStateReporter
        if storage is None:
            self._reporter = None
            return
        rep = MultiStateReporter(storage) if isinstance(storage, (str, bytes, os.PathLike)) else storage
        self._reporter = rep if hasattr(rep, 'write_iteration') else None
        if self._reporter is None or not isinstance(rep, MultiStateReporter):
            return
        # multistatesampler.py:588: never write over an existing simulation.  Rank 0 owns the storage, but EVERY rank must raise:
        # the ranks that did not would walk into the collectives create() runs next and hang there
        exists = self._comm.broadcast_object(bool(rep.storage_exists()) if self._comm.rank == 0 else None)
        if exists:
            raise RuntimeError('Storage file {} already exists; cowardly refusing to overwrite.'.format(rep.filepath))
        if self._comm.rank != 0:
            return
        if not rep.is_open() or rep._open_mode == 'r':
            if getattr(rep, 'layout', None) == 'auto' and str(rep.filepath).endswith('.nc'):
                from ._reference_store import ReferenceStoreWriter
                what = ReferenceStoreWriter.can_store(self._thermodynamic_states, self._unsampled_states, self._mcmc_moves)
                if what is not None:
                    logger.warning('%s cannot be held by the reference\'s netCDF4 layout: %s is written as a record-file container',
                                   what, rep.filepath)
                    rep.layout = 'records'
            rep.open('w')
        rep.initialize(self.n_replicas, self.n_states, len(self._unsampled_states), self._thermodynamic_states[0].n_particles)
        rep.write_thermodynamic_states(self._thermodynamic_states, self._unsampled_states)
        rep.write_mcmc_moves(self._mcmc_moves)
        rep.write_dict('options', self._options())
        rep.write_dict('metadata', self._metadata)

    _TITLE_TEMPLATE = 'Multi-state sampler simulation created using MultiStateSampler class of openmmtools_amd.multistate on {}'   # :534

    # multistatesampler.py:440-517 _StoredProperty: options that are kept in sync with the storage -- assigning one on a created
    # sampler rewrites the stored 'options' (so that a resume sees the new number of iterations, analysis interval ...)
    _STORED_OPTIONS = frozenset(['number_of_iterations', 'online_analysis_interval', 'online_analysis_target_error',
                                 'online_analysis_minimum_iterations', 'locality', 'replica_mixing_scheme', 'log_target_probabilities',
                                 'state_update_scheme', 'update_stages', 'flatness_criteria', 'flatness_threshold',
                                 'weight_update_method', 'gamma0', 'logZ_guess'])

    def __setattr__(self, name, value):
        object.__setattr__(self, name, value)
        if name in self._STORED_OPTIONS:
            d = self.__dict__
            rep = d.get('_reporter')
            if rep is not None and d.get('_thermodynamic_states') is not None and getattr(rep, '_open_mode', None) in ('w', 'a') \
                    and d.get('_comm') is not None and d['_comm'].rank == 0 and not d.get('_restoring', False):
                try:
                    rep.write_dict('options', self._options())
                except AttributeError:
                    pass                                  # (an option assigned while the constructor is still running)

    def _options(self):
        """What from_storage needs to rebuild the sampler (multistatesampler.py:1145-1167 _store_options)."""
        kwargs = dict(locality=self.locality, online_analysis_interval=self.online_analysis_interval,
                      online_analysis_target_error=self.online_analysis_target_error,
                      online_analysis_minimum_iterations=self.online_analysis_minimum_iterations)
        kwargs.update(self._ctor_kwargs())
        return dict(cls=type(self).__name__, module=type(self).__module__, number_of_iterations=self.number_of_iterations,
                    seed=self._seed, kwargs=kwargs)

    def _ctor_kwargs(self):
        return {}

    @classmethod
    def from_storage(cls, storage, engine=None, comm=None, continue_in=None, seed=None):
        """multistatesampler.py:263-299 + _restore_sampler_from_reporter (:956-1047): resume from the last checkpoint
        iteration whose data is complete.

        ``storage`` may also be a netCDF4 store WRITTEN BY THE REFERENCE (opened read-only through libhdf5, _reference_store.py):
        states, moves, options, the last checkpoint (zero velocities for legacy files, as tests/test_sampling.py:2975-2983
        asserts), energies and statistics are restored as the reference restores them and the run continues on the device.
        The reference's file is never written: ``continue_in`` names a new store of this package's format that receives the
        history read so far and every further iteration (None: the resumed sampler reports nowhere); ``seed`` is this
        engine's Philox key (the reference's random streams cannot be continued)."""
        import importlib
        from .multistatereporter import MultiStateReporter
        rep = MultiStateReporter(storage) if isinstance(storage, (str, bytes, os.PathLike)) else storage
        if not rep.is_open():
            if not rep.storage_exists():                                        # :1163-1166
                raise FileNotFoundError('Storage file {} or its subfiles do not exist; cannot read status.'.format(rep.filepath))
            rep.open('a')
        it = rep.read_last_iteration(last_checkpoint=True)
        if it is None:
            raise IOError('storage {} holds no complete checkpoint'.format(rep.filepath))
        if rep.is_reference_store:
            return cls._from_reference_store(rep, it, engine, comm, continue_in, seed)
        opts = rep.read_dict('options')
        if str(opts['module']).startswith('openmmtools_amd.'):
            klass = getattr(importlib.import_module(opts['module']), opts['cls'])
        elif cls.__module__ == opts['module'] and cls.__name__ == opts['cls']:
            klass = cls                  # a user's subclass resumes through itself: nothing named in the storage is imported
        else:
            raise TypeError('storage was written by {}.{}: resume with that class\'s own from_storage'.format(opts['module'], opts['cls']))
        if not issubclass(klass, cls):
            raise TypeError('storage was written by {}, not a {}'.format(opts['cls'], cls.__name__))
        moves = rep.read_mcmc_moves()
        s = klass(mcmc_moves=moves, number_of_iterations=opts['number_of_iterations'], engine=engine, seed=opts['seed'],
                  comm=comm, **opts['kwargs'])
        thermo, unsampled = rep.read_thermodynamic_states()
        sampler_states = rep.read_sampler_states(it)
        labels = rep.read_replica_thermodynamic_states(it)
        s._pre_write_create(thermo, sampler_states, None, initial_thermodynamic_states=labels,
                            unsampled_thermodynamic_states=unsampled, metadata=rep.read_dict('metadata'))
        s._mcmc_moves = moves
        s._iteration = int(it)
        e, nb, eu = rep.read_energies(it)
        s._energy_thermodynamic_states[:, :] = e
        s._neighborhoods[:, :] = nb
        s._energy_unsampled_states[:, :] = eu
        acc, prop = rep.read_mixing_statistics(it)
        s._n_accepted_matrix[:, :] = acc
        s._n_proposed_matrix[:, :] = prop
        s._restore_online(rep.read_online_data_if_present(it))
        if s.online_analysis_interval is not None:                              # :991-995
            last_f_k, last_free_energy = cls._read_last_free_energy(rep, it)
            if last_f_k is not None:
                s._last_mbar_f_k = np.array(last_f_k, dtype=np.float64)
                s._last_err_free_energy = float(last_free_energy[1])
        s._reporter = rep
        s._initialize_engine()
        s._mix_from_stored_energies = True
        return s

    @classmethod
    def _from_reference_store(cls, rep, it, engine, comm, continue_in, seed):
        import inspect
        from .multistatereporter import MultiStateReporter
        opts = dict(rep.read_dict('options'))
        accepted = set()
        for klass in cls.__mro__:
            if klass is object:
                continue
            accepted |= set(inspect.signature(klass.__init__).parameters)
        kwargs = {k: v for k, v in opts.items() if k in accepted and k not in ('mcmc_moves', 'number_of_iterations')}
        dropped = sorted(k for k in opts if k not in accepted)
        if dropped:
            logger.warning('options of the reference store not understood by %s and ignored: %s', cls.__name__, dropped)
        kwargs.pop('seed', None)
        moves = rep.read_mcmc_moves()
        n_iter = opts.get('number_of_iterations', 1)
        if seed is None and getattr(rep, '_ref', None) is not None:
            seed = rep._ref.read_seed()              # a store of the reference's layout written by this package carries its seed
        s = cls(mcmc_moves=moves, number_of_iterations=float('inf') if n_iter is None else n_iter, engine=engine,
                seed=0xC0FFEE if seed is None else seed, comm=comm, **kwargs)
        thermo, unsampled = rep.read_thermodynamic_states()
        sampler_states = rep.read_sampler_states(it)
        labels = np.asarray(rep.read_replica_thermodynamic_states(it), dtype=np.int64)
        s._pre_write_create(thermo, sampler_states, None, initial_thermodynamic_states=labels,
                            unsampled_thermodynamic_states=unsampled, metadata=rep.read_dict('metadata'))
        s._mcmc_moves = moves
        s._iteration = int(it)
        e, nb, eu = rep.read_energies(it)
        s._energy_thermodynamic_states[:, :] = e
        s._neighborhoods[:, :] = nb
        s._energy_unsampled_states[:, :] = eu
        acc, prop = rep.read_mixing_statistics(it)
        s._n_accepted_matrix[:, :] = acc
        s._n_proposed_matrix[:, :] = prop
        s._restore_online(rep.read_online_data_if_present(it))
        s._iteration0_energies_reported = True
        s._reporter = None
        s._initialize_engine()
        s._mix_from_stored_energies = True
        if continue_in is None:
            from ._reference_store import ReferenceStoreWriter
            if ReferenceStoreWriter.written_here(rep.filepath):
                # this package's own store in the reference's layout: extended in place (rank 0 writes)
                rep.close()
                if s._comm.rank == 0:
                    rep.open('a')
                s._reporter = rep
        if continue_in is not None:
            new = MultiStateReporter(continue_in) if isinstance(continue_in, (str, bytes, os.PathLike)) else continue_in
            if s._comm.broadcast_object(bool(new.storage_exists()) if s._comm.rank == 0 else None):
                raise RuntimeError('Storage file {} already exists; cowardly refusing to overwrite.'.format(new.filepath))
            new._checkpoint_interval = rep.checkpoint_interval
            if s._comm.rank == 0:
                new.open('w')
                new.initialize(s.n_replicas, s.n_states, len(unsampled), thermo[0].n_particles)
                new.write_thermodynamic_states(thermo, unsampled)
                new.write_mcmc_moves(moves)
                new.write_dict('options', s._options())
                new.write_dict('metadata', s._metadata)
                E, NB, EU = rep.read_energies(slice(0, it + 1))
                ST = rep.read_replica_thermodynamic_states(slice(0, it + 1))
                ACC, PROP = rep.read_mixing_statistics(slice(0, it + 1))
                for k in range(it + 1):                                  # the history the reference wrote, iteration by iteration
                    new.write_replica_thermodynamic_states(ST[k], k)
                    new.write_energies(E[k], NB[k], EU[k], k)
                    new.write_mixing_statistics(ACC[k], PROP[k], k)
                    new.write_timestamp(k)
                    ck = rep.read_sampler_states(k)
                    if ck is not None:
                        new.write_sampler_states(ck, k)
                new.write_last_iteration(it)
            s._reporter = new
        return s

    def _restore_online(self, data):
        pass

    def _gather_sampler_states(self):
        """Checkpoint read point (multistatesampler.py:1217): device -> host for the local block, then (multi-rank) a
        host gather of the blocks to rank 0.  Collective: every rank calls it on checkpoint iterations."""
        self._sync_sampler_states()
        if self._comm.world_size == 1:
            return
        mine = [(s.positions, s.velocities) for s in self._sampler_states[self._r_begin:self._r_begin + self._r_count]]
        blocks = self._comm.gather_objects((self._r_begin, mine))
        if blocks is not None:
            for begin, block in blocks:
                for k, (x, v) in enumerate(block):
                    self._sampler_states[begin + k].positions = x
                    self._sampler_states[begin + k].velocities = v

    def _pre_write_create(self, thermodynamic_states, sampler_states, storage, initial_thermodynamic_states=None,
                          unsampled_thermodynamic_states=None, metadata=None):
        """multistatesampler.py:836-926."""
        is_periodic = thermodynamic_states[0].is_periodic
        for ts in thermodynamic_states:
            if ts.is_periodic != is_periodic:
                raise Exception('Thermodynamic states contain a mixture of systems with and without '
                                'periodic boundary conditions.')
        if is_periodic:
            for ss in sampler_states:
                if ss.box_vectors is None:
                    raise Exception('All sampler states must have box_vectors defined if the system is periodic.')
        n_particles = thermodynamic_states[0].n_particles
        for the_states in (thermodynamic_states, sampler_states):
            for state in the_states:
                if state.n_particles != n_particles:
                    raise ValueError('All ThermodynamicStates and SamplerStates must have the same number '
                                     'of particles')
        # multistatesampler.py:870-877: a default title unless the caller's metadata carries one
        self._metadata = dict(metadata) if metadata else {}
        if 'title' not in self._metadata:
            self._metadata['title'] = self._TITLE_TEMPLATE.format(time.asctime(time.localtime()))
        self._thermodynamic_states = copy.deepcopy(list(thermodynamic_states))
        self._unsampled_states = copy.deepcopy(list(unsampled_thermodynamic_states or []))
        self._sampler_states = [copy.deepcopy(s) for s in sampler_states]
        if initial_thermodynamic_states is None:
            initial_thermodynamic_states = self._default_initial_thermodynamic_states(thermodynamic_states,
                                                                                      sampler_states)
        self._replica_thermodynamic_states = np.array(initial_thermodynamic_states, np.int64)
        for replica_id, state_id in enumerate(self._replica_thermodynamic_states):
            ss = self._sampler_states[replica_id]
            if ss.box_vectors is None:
                ss.box_vectors = self._thermodynamic_states[state_id].system.getDefaultPeriodicBoxVectors()
        if isinstance(self._mcmc_moves, mcmc.MCMCMove):
            self._mcmc_moves = [copy.deepcopy(self._mcmc_moves) for _ in range(self.n_states)]
        elif len(self._mcmc_moves) != self.n_states:
            raise RuntimeError('The number of MCMCMoves ({}) and ThermodynamicStates ({}) must be the same.'.format(
                len(self._mcmc_moves), self.n_states))
        self._iteration = 0
        K, R = self.n_states, self.n_replicas
        self._n_accepted_matrix = np.zeros([K, K], np.int64)
        self._n_proposed_matrix = np.zeros([K, K], np.int64)
        self._energy_thermodynamic_states = np.zeros([R, K], np.float64)
        self._neighborhoods = np.zeros([R, K], 'i1')
        self._energy_unsampled_states = np.zeros([R, len(self._unsampled_states)], np.float64)

    @staticmethod
    def _flatten(move):
        if isinstance(move, mcmc.SequenceMove):
            out = []
            for m in move.move_list:
                out.extend(MultiStateSampler._flatten(m))
            return out
        return [move]

    @staticmethod
    def _move_key(m):
        if isinstance(m, mcmc.MonteCarloBarostatMove):
            return ('barostat', m.n_attempts)
        if isinstance(m, mcmc.MetropolizedMove):
            sub = m.atom_subset
            sub = None if sub is None else ((sub.start, sub.stop, sub.step) if isinstance(sub, slice) else tuple(int(i) for i in sub))
            return ('metropolized', type(m).__name__, sub, getattr(m, 'displacement_sigma', None))
        if isinstance(m, mcmc.LangevinSplittingDynamicsMove):
            return ('langevin', m.timestep, m.collision_rate, m.n_steps, m.reassign_velocities, m.splitting,
                    getattr(m, 'measure_heat', False), getattr(m, 'measure_shadow_work', False))
        raise NotImplementedError('the device engine propagates with Langevin(Splitting)DynamicsMove, GHMCMove, HMCMove, '
                                  'MonteCarloBarostatMove and the Metropolized displacement / rotation moves (alone or in a SequenceMove) only')

    def _engine_program(self):
        """The per-iteration recipe of every replica: the flattened move sequence of state 0, which all states must share
        (one batched launch covers every replica).  The engine holds one integrator program at a time: a sequence with several
        integrator moves (tests/test_mcmc.py:283 SequenceMove([LangevinDynamicsMove, GHMCMove])) reprograms it between them."""
        prog = self._flatten(self._mcmc_moves[0])
        keys = [self._move_key(m) for m in prog]
        for other in self._mcmc_moves[1:]:
            if [self._move_key(m) for m in self._flatten(other)] != keys:
                raise NotImplementedError('per-state MCMC moves must be identical for batched propagation')
        return prog

    def _engine_move(self):
        """The Langevin move of the program (None for a barostat-only recipe)."""
        for m in self._engine_program():
            if isinstance(m, mcmc.LangevinSplittingDynamicsMove):
                return m
        return None

    def _program_engine_move(self, move=None):
        move = self._engine_move() if move is None else move
        eng = self._engine
        if move is not None:
            has_constraints = self._thermodynamic_states[0].system.getNumConstraints() > 0
            if getattr(eng, 'is_device', False) and has_constraints and move.constraint_tolerance < 2e-7 and not getattr(self, '_warned_tolerance', False):
                # include/remd_hip.h (remd_get_constraint_stats): the fp32 state bounds what the X-H solver can reach (SETTLE waters are analytic)
                logger.warning('constraint_tolerance %g is below what fp32 coordinates can hold: the device iterates X-H clusters to a '
                               'relative bond-length error of 2e-7 (at most 8 Newton updates) and solves rigid waters analytically',
                               move.constraint_tolerance)
                self._warned_tolerance = True
            eng.set_integrator(move.splitting, getattr(move, 'engine_timestep', move.timestep), move.collision_rate, move.n_steps,
                               move.reassign_velocities, move.constraint_tolerance)
            eng.set_restart_attempts(getattr(move, 'n_restart_attempts', 0))      # mcmc.py:706-759
            if hasattr(eng, 'set_work_measurement'):                              # mcmc.py:1308-1316: the move's flags reach the integrator
                eng.set_work_measurement(getattr(move, 'measure_heat', False), getattr(move, 'measure_shadow_work', False))

    def _state_energy_constants(self, states):
        """Additive per-state potential constants: the lambda-dependent long-range correction of the
        alchemical CustomNonbondedForces (alchemy.py:1786-1789), constant in NVT."""
        from ..system import NonbondedForce
        from ..alchemy import alchemical_long_range_constants
        system = states[0].system
        regions = getattr(system, 'alchemical_regions', None)
        if getattr(system, 'alchemical_region', None) is None and regions is None:
            return np.zeros(len(states))
        nb = [f for f in system.getForces() if isinstance(f, NonbondedForce)]
        if not nb or not nb[0].usesPeriodicBoundaryConditions():
            return np.zeros(len(states))
        volume = self._sampler_states[0].volume
        if regions is not None:               # general regions: one lambda_sterics per region and state
            lam = [s.region_lambdas([r.name for r in regions])[0] for s in states]
            return alchemical_long_range_constants(system, nb[0], lam, volume)
        return alchemical_long_range_constants(system, nb[0], [s.lambda_sterics for s in states], volume)

    def _initialize_engine(self):
        if self._engine is None:
            from .._engine import HipEngine
            # raises if libremd_hip.so / a GPU is missing: no CPU fallback.  A ContextCache assigned the reference's way
            # (sampler_context_cache, multistatesampler.py:174-175) still chooses the device
            cc = self.sampler_context_cache
            self._engine = HipEngine(device=cc.device_index) if hasattr(cc, 'device_index') else HipEngine()
        all_states = list(self._thermodynamic_states) + list(self._unsampled_states)
        ref = all_states[0]
        box0 = self._sampler_states[0].box_edges if ref.is_periodic else None
        split = getattr(self._engine, 'ewald_split', None)        # the engine's preferred split of the Ewald sum (PME systems)
        min_edge = None
        if ref.is_periodic and split == 'auto':
            # ONE Ewald split (Coulomb range, alpha, mesh) serves the whole ensemble: choose it for the SMALLEST box any replica
            # starts in, and under a barostat leave the volume room to fluctuate -- the stretched Coulomb range must stay below
            # half the box edge of every replica at every step (the engine refuses a smaller box: remd_set_replicas, and the
            # Monte Carlo barostat raises when a trial box gets there, as OpenMM does) -- ADVICE r4
            edges = np.array([np.asarray(ss.box_edges, dtype=np.float64) for ss in self._sampler_states if ss.box_edges is not None])
            min_edge = float(edges.min()) if len(edges) else float(np.min(box0))
            if any(s.pressure is not None for s in all_states):
                min_edge /= 1.1
        # states.py:186-217: states of one standard System share a handle; several Systems => one handle per group behind the
        # same interface (_engine_pool.py), as the reference keeps one Context per compatible group (multistatesampler.py:1470-1490)
        from ..states import group_by_compatibility
        groups, group_indices = group_by_compatibility(all_states)
        if len(groups) > 1:
            if any(getattr(g[0].system, 'alchemical_region', None) is not None or getattr(g[0].system, 'alchemical_regions', None) is not None for g in groups):
                raise NotImplementedError('alchemical states in more than one compatibility group')
            from ._engine_pool import EnginePool
            if not isinstance(self._engine, EnginePool):
                self._engine = EnginePool(self._engine, group_indices)
            desc = [system_to_desc(g[0].system, box=box0, ewald_split=split, min_edge=min_edge) for g in groups]
        else:
            desc = system_to_desc(ref.system, box=box0, ewald_split=split, min_edge=min_edge)
        eng = self._engine
        eng.set_system(desc)
        beta = np.array([s.beta for s in all_states])
        lam_s = np.array([s.lambda_sterics for s in all_states], dtype=np.float64)
        lam_e = np.array([s.lambda_electrostatics for s in all_states], dtype=np.float64)
        eng.set_states(beta, lam_s, lam_e, self._state_energy_constants(all_states))
        regions = getattr(ref.system, 'alchemical_regions', None)
        if regions is not None:
            # general alchemical regions (alchemy.py here; csrc/alch_regions.hip): every region's lambdas at every state
            names = [r.name for r in regions]
            lams = [s.region_lambdas(names) for s in all_states]
            eng.set_region_lambdas(np.array([l[0] for l in lams], dtype=np.float64), np.array([l[1] for l in lams], dtype=np.float64))
            bonded = np.array([s.region_bonded_lambdas(names) for s in all_states], dtype=np.float64)           # [K][3][n]
            if np.any(bonded != 1.0):
                eng.set_region_bonded_lambdas(bonded[:, 0], bonded[:, 1], bonded[:, 2])
        elif any(getattr(s, k, 1.0) != 1.0 for s in all_states for k in ('lambda_bonds', 'lambda_angles', 'lambda_torsions')):
            raise NotImplementedError('lambda_bonds / lambda_angles / lambda_torsions act on the bonded terms an AlchemicalRegion names '
                                      '(alchemical_bonds=..., alchemical_angles=..., alchemical_torsions=...): this System has none')
        self._program_engine_move()
        pressures = [s.pressure for s in all_states]
        if any(p is not None for p in pressures):
            if any(p is None for p in pressures):
                raise ValueError('NPT and NVT thermodynamic states cannot be mixed')
            eng.set_barostat(np.array(pressures, dtype=np.float64), all_states[0].barostat_frequency)
            # the alchemical long-range constants were evaluated at the first sampler state's volume and scale as 1/V
            eng.set_energy_const_volume(self._sampler_states[0].volume if np.any(self._state_energy_constants(all_states) != 0.0) else 0.0)
            self._npt = True
        else:
            eng.set_barostat(None)
            eng.set_energy_const_volume(0.0)
            self._npt = False
        eng.seed(self._seed)
        R = self.n_replicas
        self._r_begin, self._r_count = self._comm.partition(R)
        sl = slice(self._r_begin, self._r_begin + self._r_count)
        x = np.stack([s.positions for s in self._sampler_states[sl]])
        have_v = all(s.velocities is not None for s in self._sampler_states[sl])
        v = np.stack([s.velocities for s in self._sampler_states[sl]]) if have_v else None
        if ref.is_periodic:
            box = np.stack([s.box_edges for s in self._sampler_states[sl]])
        else:
            box = np.zeros((self._r_count, 3))
        eng.set_replicas(R, self._r_begin, x, v, box, self._replica_thermodynamic_states)
        self._K_total = len(all_states)

    # ---- run / equilibrate ------------------------------------------------------------------
    @with_timer('Minimizing all replicas')
    def minimize(self, tolerance=1.0 * unit.kilojoules_per_mole / unit.nanometers, max_iterations=0):
        """multistatesampler.py:611-647: FIRE-minimise every replica at its current state (one batched device call per
        rank instead of one Context per replica), store the minimised positions in the sampler states and in storage."""
        if self._thermodynamic_states is None or self.n_replicas == 0:
            raise RuntimeError('Cannot minimize replicas. The simulation must be created first.')     # :629-630
        self._engine.set_labels(self._replica_thermodynamic_states)
        converged, n_steps = self._engine.minimize(float(unit.to_md(tolerance)), int(max_iterations))
        self._sampler_states_stale = True
        self._neighborhoods[:, :] = 0                     # energies are stale: run() recomputes them at iteration 0
        self._gather_sampler_states()
        if self._reporter is not None and hasattr(self._reporter, 'write_sampler_states') and self._comm.rank == 0 \
                and getattr(self._reporter, '_meta', None) is not None:
            self._reporter.write_sampler_states(self._sampler_states, self._iteration)                 # :647
        return converged, n_steps

    def equilibrate(self, n_iterations, mcmc_moves=None):
        """multistatesampler.py:649-722: propagate -> energies -> mix, iteration counter untouched."""
        if self._thermodynamic_states is None:
            raise RuntimeError('Cannot equilibrate replicas. The simulation must be created first.')
        production_moves = self._mcmc_moves
        if mcmc_moves is not None:                                       # :655-670: temporary equilibration moves
            if isinstance(mcmc_moves, mcmc.MCMCMove):
                mcmc_moves = [copy.deepcopy(mcmc_moves) for _ in range(self.n_states)]
            elif len(mcmc_moves) != self.n_states:
                raise RuntimeError('The number of MCMCMoves ({}) and ThermodynamicStates ({}) for equilibration'
                                   ' must be the same.'.format(len(mcmc_moves), self.n_states))
            self._mcmc_moves = list(mcmc_moves)
            try:
                self._program_engine_move()
            except Exception:
                self._mcmc_moves = production_moves
                raise
        try:
            if self._iteration == 0 and not self._energies_computed():
                self._compute_energies()
            for it in range(1, 1 + n_iterations):
                self._equil_iteration = it
                self._propagate_replicas(rng_iteration=-it)
                self._compute_energies()
                self._replica_thermodynamic_states = self._mix_replicas(rng_iteration=-it)
            self._check_nan_energy()
        finally:
            if self._mcmc_moves is not production_moves:                 # :716-717: restore the production moves
                self._mcmc_moves = production_moves
                self._program_engine_move()
        if self._reporter is not None:                                   # :720-721: update the stored positions
            self._gather_sampler_states()
            if self._comm.rank == 0:
                self._reporter.write_sampler_states(self._sampler_states, self._iteration)

    def _energies_computed(self):
        return bool(self._neighborhoods.any())

    def run(self, n_iterations=None):
        """multistatesampler.py:724-804."""
        if self._thermodynamic_states is None:
            raise RuntimeError('call create() first')
        if self._iteration == 0 and not getattr(self, '_iteration0_energies_reported', False):
            # :738-753: at iteration 0 the starting energies (of the minimised / equilibrated structures) are ALWAYS
            # computed and written, whatever equilibrate() or minimize() did before
            if not self._energies_computed():
                self._compute_energies()
            self._check_nan_energy()
            if self._reporter is not None:
                self._reporter.write_iteration(self)                   # iteration 0 again, now with the starting energies
            self._iteration0_energies_reported = True
        if n_iterations is None:
            iteration_limit = self.number_of_iterations
        else:
            iteration_limit = min(self._iteration + n_iterations, self.number_of_iterations)
        t_run = time.time()
        run_initial_iteration = self._iteration
        while not self._is_completed(iteration_limit):                 # :766
            t0 = time.time()
            self._iteration += 1                                       # :768
            logger.info('Iteration %d/%s', self._iteration, iteration_limit)
            self._replica_thermodynamic_states = self._mix_replicas()  # :776
            t1 = time.time()
            self._propagate_replicas()                                 # :779
            t2 = time.time()
            self._compute_energies()                                   # :782
            t3 = time.time()
            self._report_iteration()                                   # :785
            self._update_analysis()                                    # :788
            self._update_timing(t0, t1, t2, t3, t_run, run_initial_iteration, iteration_limit)   # :793
            logger.info('Iteration took %.3fs.', self._timing_data['iteration_seconds'])
            if 'estimated_time_remaining' in self._timing_data:
                logger.info('Estimated completion in %s, at %s (consuming total wall clock time %s).',
                            self._timing_data['estimated_time_remaining'], self._timing_data['estimated_localtime_finish_date'],
                            self._timing_data['estimated_total_time'])
            self._check_nan_energy()                                   # :804

    def _update_timing(self, t0, t1, t2, t3, t_run, run_initial_iteration=None, iteration_limit=None):
        """multistatesampler.py:1766-1803: per-iteration and average wall time, completion estimate, ns/day over all
        replicas' dynamic moves (plus the mix / propagate / energy split of the iteration, which the reference only logs)."""
        import datetime
        d = self._timing_data
        d['iteration_seconds'] = t3 - t0
        d['mixing_seconds'] = t1 - t0
        d['propagation_seconds'] = t2 - t1
        d['energy_seconds'] = t3 - t2
        n = (self._iteration - run_initial_iteration) if run_initial_iteration is not None else d.get('n_timed', 0) + 1
        d['n_timed'] = n
        d['average_seconds_per_iteration'] = (time.time() - t_run) / n if n > 0 else 0.0
        if iteration_limit is not None and np.isfinite(iteration_limit):
            remaining = datetime.timedelta(seconds=d['average_seconds_per_iteration'] * (iteration_limit - self._iteration))
            d['estimated_time_remaining'] = str(remaining)
            d['estimated_localtime_finish_date'] = (datetime.datetime.now() + remaining).strftime('%Y-%b-%d-%H:%M:%S')
            d['estimated_total_time'] = str(datetime.timedelta(seconds=d['average_seconds_per_iteration'] * iteration_limit))
        ns_per_iter = 0.0
        for state_move in self._mcmc_moves:                                # one (possibly composite) move per replica
            for move in self._flatten(state_move):
                if hasattr(move, 'timestep') and hasattr(move, 'n_steps'):
                    ns_per_iter += move.timestep * 1e-3 * move.n_steps      # timestep in ps
        avg = d['average_seconds_per_iteration']
        d['ns_per_day'] = ns_per_iter / (avg / 86400.0) if avg > 0 else 0.0

    @with_timer('Writing iteration information to storage')
    def _report_iteration(self):
        if self._reporter is not None:
            self._reporter.write_iteration(self)

    # ---- completion and online / offline analysis (multistatesampler.py:526-528, 1519-1735) ----------------------
    @property
    def is_completed(self):
        return self._is_completed()

    def _is_completed(self, iteration_limit=None):
        if iteration_limit is None:
            iteration_limit = self.number_of_iterations
        return self._is_completed_static(iteration_limit, self._iteration, self._last_err_free_energy,
                                         self.online_analysis_target_error)

    @staticmethod
    def _is_completed_static(iteration_limit, iteration, last_err_free_energy, online_analysis_target_error):
        """:1720-1730: the iteration limit, or the statistical error target of the online analysis."""
        return bool(iteration >= iteration_limit or (last_err_free_energy is not None and
                                                     last_err_free_energy <= online_analysis_target_error))

    def _update_analysis(self):
        """:1677-1694: the stochastic-approximation estimate every iteration, MBAR every online_analysis_interval."""
        if self.online_analysis_interval is None:
            return
        self._last_err_free_energy = self._online_analysis()
        if self._iteration % self.online_analysis_interval == 0:
            err = self._offline_analysis()
            if err is not None:
                self._last_err_free_energy = err

    @with_timer('Computing online free energy estimate')
    def _online_analysis(self, gamma0=1.0):
        """:1625-1675.  logZ_k += gamma * P(k | x_r) / pi_k over the replicas (pi_k = 1, gamma = gamma0 / (iteration + 1)),
        then anchored at state 0.  Replicated on every rank from the replicated u_kl (the reference: rank 0 + broadcast)."""
        gamma = gamma0 / float(self._iteration + 1)
        if self._last_mbar_f_k is None:
            self._last_mbar_f_k = np.zeros([self.n_states], np.float64)
        logZ = -self._last_mbar_f_k
        u = self._energy_thermodynamic_states
        if self.locality is None:
            # log P_k = -u_k - logsumexp(-u): global neighbourhoods, zero log weights (:1647-1653)
            m = np.max(-u, axis=1, keepdims=True)
            log_P = -u - (m + np.log(np.sum(np.exp(-u - m), axis=1, keepdims=True)))
            P = np.exp(log_P)
            for r in range(self.n_replicas):                 # sequential accumulation, the reference's order
                logZ += gamma * P[r]
        else:
            for r, state in enumerate(self._replica_thermodynamic_states):
                nb = self._neighborhood(state)
                lp = -u[r, nb]
                lp = lp - (lp.max() + np.log(np.sum(np.exp(lp - lp.max()))))
                logZ[nb] += gamma * np.exp(lp)
        logZ[:] -= logZ[0]
        self._last_mbar_f_k = -logZ
        free_energy = self._last_mbar_f_k[-1] - self._last_mbar_f_k[0]
        self._last_err_free_energy = np.inf
        if self._reporter is not None and self._comm.rank == 0:
            self._reporter.write_online_data_dynamic_and_static(self._iteration, f_k=self._last_mbar_f_k,
                                                                free_energy=(free_energy, self._last_err_free_energy))
        return self._last_err_free_energy

    @with_timer('Computing offline free energy estimate')
    def _offline_analysis(self):
        """:1522-1620: MBAR on the stored, equilibrated and decorrelated energies; needs a reporter.  Returns the standard
        error of f_last - f_first in kT, or None when the estimate cannot be computed yet (under-sampled)."""
        if self.locality is not None:
            raise Exception('Cannot use MBAR with non-global locality.')
        if self._reporter is None:
            return None
        from .analysis import MultiStateSamplerAnalyzer
        if not hasattr(self, '_last_mbar_f_k_offline'):
            self._last_mbar_f_k_offline = np.zeros(self.n_states + len(self._unsampled_states))
        err = None
        if self._comm.rank == 0:
            analysis = MultiStateSamplerAnalyzer(self._reporter, analysis_kwargs={'initial_f_k': self._last_mbar_f_k_offline})
            try:
                mbar = analysis.mbar
                free_energy, err_free_energy = analysis.get_free_energy()
                n_eq, g_t = analysis.n_equilibration_iterations, analysis.statistical_inefficiency
            except Exception as e:            # under-sampled data must not stop a running simulation (:1566-1575 traps pymbar's ParameterError)
                logger.debug('MBAR could not be computed: %s: %s', type(e).__name__, e)
            else:
                self._last_mbar_f_k_offline = mbar.f_k
                fe, err = float(free_energy[0, -1]), float(err_free_energy[0, -1])
                if np.isnan(err):
                    err = np.inf                                            # :1585-1586
                self._reporter.write_online_data_dynamic_and_static(self._iteration, f_k_offline=self._last_mbar_f_k_offline,
                                                                    free_energy=(fe, err))
                self._reporter.write_current_statistics({
                    'iteration': int(self._iteration),
                    'percent_complete': float(self._iteration * 100 / self.number_of_iterations),
                    'mbar_analysis': {'free_energy_in_kT': fe, 'standard_error_in_kT': float(err),
                                      'number_of_uncorrelated_samples': float(analysis._equilibration_data[-1]),
                                      'n_equilibrium_iterations': int(n_eq), 'statistical_inefficiency': float(g_t)},
                    'timing_data': {k: (v if isinstance(v, str) else float(v)) for k, v in self._timing_data.items()}})
        if self._comm.world_size > 1:
            err = self._comm.broadcast_object(err)
        return err

    @staticmethod
    def _read_last_free_energy(reporter, iteration):
        """:1696-1718: the most recent stored (f_k, (free_energy, error)) at or before ``iteration``."""
        last_f_k = last_free_energy = None
        for index in range(iteration, 0, -1):
            try:
                data = reporter.read_online_analysis_data(index, 'f_k', 'free_energy')
                last_f_k, last_free_energy = data['f_k'], data['free_energy']
            except (IndexError, KeyError, ValueError):
                break
            if not np.all(last_f_k == 0):
                break
        return last_f_k, last_free_energy

    # ---- the three hooks ---------------------------------------------------------------------
    def _mix_replicas(self, rng_iteration=None):
        """multistatesampler.py:1500-1517: the base class does not mix."""
        self._n_accepted_matrix[:, :] = 0
        self._n_proposed_matrix[:, :] = 0
        return self._replica_thermodynamic_states

    @with_timer('Propagating all replicas')
    def _propagate_replicas(self, rng_iteration=None):
        """multistatesampler.py:1287-1337 for every local replica in one device call."""
        it = self._iteration if rng_iteration is None else rng_iteration
        self._engine.set_labels(self._replica_thermodynamic_states)
        program = self._engine_program()                           # SequenceMove: in order, once per iteration (mcmc.py:406-424)
        integrations = [m for m in program if isinstance(m, mcmc.LangevinSplittingDynamicsMove)]
        for position, move in enumerate(program):
            if isinstance(move, mcmc.MonteCarloBarostatMove):
                if not getattr(self, '_npt', False):
                    raise RuntimeError('Requested a MonteCarloBarostat move' ' on a system at constant pressure')   # mcmc.py:1673-1676 (the reference's wording: it means a state without a barostat)
                self._engine.barostat_attempts(move.n_attempts)
                self._sampler_states_stale = True
                continue
            if isinstance(move, mcmc.MetropolizedMove):
                self._apply_metropolized_move(position, move, it)
                self._sampler_states_stale = True
                continue
            key = it
            if len(integrations) > 1:
                # one integrator program at a time: load this move's; its noise is keyed by (iteration, place in the sequence).
                # The place rides in bits 34.. of the iteration (below the restart attempts' bit 40): the engine numbers the
                # steps of a propagation key * n_steps + step with the CURRENT move's n_steps, so a key scaled by the number of
                # moves let the step ranges of moves with different n_steps overlap
                self._program_engine_move(move)
                nth = sum(1 for m in program[:position] if isinstance(m, mcmc.LangevinSplittingDynamicsMove))
                if nth >= 64:
                    raise NotImplementedError('more than 64 integrator moves in one SequenceMove')
                key = it + (nth << 34) * (1 if it >= 0 else -1)
            counted = isinstance(move, mcmc.GHMCMove) and hasattr(self._engine, 'get_work')
            before = self._engine.get_work() if counted else None
            flags = self._engine.propagate(key)
            self._sampler_states_stale = True
            if np.any(flags):
                self._dump_nan_errors(np.nonzero(flags)[0], move)
            if counted:
                self._credit_metropolized_steps(position, before, self._engine.get_work())
        for state_move in self._mcmc_moves:
            for move in self._flatten(state_move):
                if not isinstance(move, (mcmc.GHMCMove, mcmc.MetropolizedMove)):
                    move.statistics = dict(n_attempts=move.statistics.get('n_attempts', 0) + 1)

    def _apply_metropolized_move(self, position, move, it):
        """mcmc.py:843-905 for every local replica at once: potential of the current positions, proposal for the move's atom
        subset, potential of the proposed positions (two batched device evaluations at each replica's own state), Metropolis
        test on beta * delta U (a rigid move of a subset leaves the volume, hence the p V term, unchanged).  Proposals and the
        acceptance draws come from one numpy Philox stream per (seed, iteration, place in the sequence, global replica): a
        run is reproducible and independent of how replicas are sharded (the reference draws from numpy's global stream)."""
        eng = self._engine
        x, v, u_old, _ = eng.get_replicas(positions=True, velocities=True, potential=True)
        n_local = x.shape[0]
        if self._thermodynamic_states[0].is_periodic:
            box = np.asarray(eng.get_boxes(), dtype=np.float64).reshape(n_local, 3)
        else:
            box = np.zeros((n_local, 3))
        subset = move._subset()
        proposed = x.copy()
        rngs = []
        for r in range(n_local):
            words = [self._seed & 0xFFFFFFFFFFFFFFFF, int(it) & 0xFFFFFFFFFFFFFFFF, int(position), int(self._r_begin + r), 0x4D43]
            rng = np.random.Generator(np.random.Philox(np.random.SeedSequence(words)))
            rngs.append(rng)
            proposed[r][subset] = move._propose_positions(x[r][subset], rng)
        labels = self._replica_thermodynamic_states
        eng.set_replicas(self.n_replicas, self._r_begin, proposed, v, box, labels)
        u_new = eng.get_replicas(positions=False, velocities=False, potential=True)[2]
        states = np.asarray(labels)[self._r_begin:self._r_begin + n_local]
        beta = np.array([self._thermodynamic_states[int(k)].beta for k in states])
        delta = beta * (np.asarray(u_new) - np.asarray(u_old))
        accept = np.zeros(n_local, dtype=bool)
        for r in range(n_local):
            accept[r] = (not np.isnan(u_new[r])) and (delta[r] <= 0.0 or rngs[r].random() < np.exp(-delta[r]))     # :893-895
            mv = self._flatten(self._mcmc_moves[int(states[r])])[position]
            mv.n_accepted += int(accept[r])
            mv.n_proposed += 1
        if not np.all(accept):
            proposed[~accept] = x[~accept]                                    # :897-898: restore the rejected replicas
            eng.set_replicas(self.n_replicas, self._r_begin, proposed, v, box, labels)

    def _credit_metropolized_steps(self, position, before, after):
        """mcmc.py:1478-1489: a GHMCMove accumulates the accepted / attempted steps of the integrations it drove.  The engine
        counts per replica; the steps of this integration go to the move (at ``position`` of the flattened sequence) of the
        state each local replica was propagated in.  Under several ranks every rank credits its own replicas, as the
        reference's worker-side move objects do."""
        acc = np.asarray(after['n_accepted'], np.int64) - np.asarray(before['n_accepted'], np.int64)
        tri = np.asarray(after['n_trials'], np.int64) - np.asarray(before['n_trials'], np.int64)
        local_states = np.asarray(self._replica_thermodynamic_states)[self._r_begin:self._r_begin + len(acc)]
        for r, state in enumerate(local_states):
            move = self._flatten(self._mcmc_moves[int(state)])[position]
            move.n_accepted += int(acc[r])
            move.n_proposed += int(tri[r])

    def _dump_nan_errors(self, bad_local, move):
        """multistatesampler.py:1324-1334: save the NaN-ing replica (state, System, move) under ``nan-error-logs/`` next to
        the storage before aborting with SimulationNaNError."""
        x, v, _, _ = self._engine.get_replicas()
        try:
            boxes = self._engine.get_boxes()
        except Exception:
            boxes = None
        base = os.path.dirname(os.path.abspath(self._reporter.filepath)) if self._reporter is not None else os.getcwd()
        output_dir = os.path.join(base, 'nan-error-logs')
        bad_global = []
        for rl in bad_local:
            replica_id = int(rl) + self._r_begin
            state_id = int(self._replica_thermodynamic_states[replica_id])
            before = self._sampler_states[replica_id]            # last state synchronised to the host (start of the iteration at best)
            err = mcmc.IntegratorMoveError('NaN after %s' % type(move).__name__, move, context=dict(
                system=self._thermodynamic_states[state_id].system, thermodynamic_state=self._thermodynamic_states[state_id],
                positions=x[rl], velocities=v[rl], box=None if boxes is None else boxes[rl],
                positions_before=before.positions, velocities_before=before.velocities))
            err.serialize_error(os.path.join(output_dir, 'iteration{}-replica{}-state{}'.format(self._iteration, replica_id, state_id)))
            bad_global.append(replica_id)
        message = ('Propagating replicas {} resulted in a NaN!\nThe state of the system and integrator before the error were saved'
                   ' in {}').format(bad_global, output_dir)
        logger.critical(message)
        raise SimulationNaNError(message)

    @with_timer('Computing energy matrix')
    def _compute_energies(self):
        """multistatesampler.py:1436-1494: fill u_kl for all replicas (global neighborhoods)."""
        K, U = self.n_states, len(self._unsampled_states)
        if isinstance(self._comm, SingleProcessComm):
            rows = self._engine.compute_energies()
            full = rows
        else:
            full = self._gather_rows()
        full = np.asarray(full)
        if U:
            self._energy_unsampled_states[:, :] = full[:, K:]
        if self.locality is None:
            self._energy_thermodynamic_states[:, :] = full[:, :K]
            self._neighborhoods[:, :] = 1                               # :1441-1444: global neighbourhoods
            return
        # local neighbourhoods (:1441-1458): only the energies of the states within +-locality of a replica's state are
        # refreshed and flagged; the device computes whole rows anyway, so this is a mask on what is kept
        self._neighborhoods[:, :] = 0
        for r, state in enumerate(self._replica_thermodynamic_states):
            nb = self._neighborhood(state)
            self._neighborhoods[r, nb] = 1
            self._energy_thermodynamic_states[r, nb] = full[r, nb]

    def _neighborhood(self, state_index):
        """:1263-1281."""
        if self.locality is None:
            return list(range(0, self.n_states))
        state_index = int(state_index)
        return list(range(max(0, state_index - self.locality), min(self.n_states, state_index + self.locality + 1)))

    def _gather_rows(self):
        """Multi-rank: each rank computes its [R_local, K_total] rows; RCCL/gloo all-gather."""
        import torch
        eng = self._engine
        Kt = self._K_total
        if getattr(eng, 'is_device', False):
            if self._device_ukl is None:
                dev = torch.device('cuda', eng.device)
                self._device_rows = torch.empty((self._r_count, Kt), dtype=torch.float64, device=dev)
                self._device_ukl = torch.empty((self.n_replicas, Kt), dtype=torch.float64, device=dev)
            eng.compute_energies(d_rows=self._device_rows.data_ptr(), want_host=False)
            t_gather = time.time()                          # (compute_energies returns synchronised: what follows is the collective)
            self._comm.all_gather_rows(self._device_rows, self.n_replicas, out=self._device_ukl)
            torch.cuda.synchronize(self._device_ukl.device)
            self._timing_data['allgather_seconds'] = time.time() - t_gather
            return self._device_ukl.cpu().numpy()
        rows = torch.from_numpy(np.ascontiguousarray(eng.compute_energies()))
        t_gather = time.time()
        full = self._comm.all_gather_rows(rows, self.n_replicas)
        self._timing_data['allgather_seconds'] = time.time() - t_gather
        self._host_ukl_full = full.numpy()
        return self._host_ukl_full

    def _check_nan_energy(self):
        """multistatesampler.py:1049-1081."""
        diag = self._energy_thermodynamic_states[np.arange(self.n_replicas), self._replica_thermodynamic_states]
        bad = np.nonzero(np.isnan(diag))[0]
        if len(bad):
            raise SimulationNaNError('NaN encountered in energies of replicas {} at their current states'.format(bad.tolist()))

    def _sync_sampler_states(self):
        """D2H snapshot of the local replicas (mcmc.py:731-773; reporter read point multistatesampler.py:1217)."""
        if not self._sampler_states_stale:
            return
        x, v, _, _ = self._engine.get_replicas(positions=True, velocities=True)
        for k in range(self._r_count):
            s = self._sampler_states[self._r_begin + k]
            s.positions = x[k].copy()
            s.velocities = v[k].copy()
        if getattr(self, '_npt', False):                                # the barostat rescales the boxes on the device
            boxes = self._engine.get_boxes()
            for k in range(self._r_count):
                self._sampler_states[self._r_begin + k].box_vectors = np.diag(boxes[k])
        self._sampler_states_stale = False

"""Storage of a multistate simulation: the write path that follows the timed iteration, and the read path
``from_storage`` / analysis need.

Mirrors openmmtools/multistate/multistatereporter.py (class :69): one *analysis* store written every
iteration — energies f8[iter, R, K] + neighborhoods i1 + unsampled f8[iter, R, U] (``write_energies``
:865-929), replica state indices i4[iter, R] (:797-815), accepted / proposed i4[iter, K, K]
(``write_mixing_statistics`` :957-999), timestamps (:1001-1018), last good iteration (:1072-1092), online
logZ / weights for SAMS (:1167-1252) — and one *checkpoint* store written every ``checkpoint_interval``
iterations (default 50, :131) with positions / velocities as **f4** and box vectors
(``_write_sampler_states_to_given_file`` :1654-1737).

The reference writes NetCDF4; netCDF4 is not available in this environment, so the container is
a directory of append-only little-endian record files (record i = iteration i, O(1) per iteration, overwritable
on resume) plus one ``.npz`` per checkpoint; variable names, dtypes and shapes are the reference's.  Only the
caller's rank 0 writes (the reference: ``@mpiplus.on_single_node(0)``).
"""
import json
import logging
import os
import pickle
import re
import time

import numpy as np


logger = logging.getLogger(__name__)


class _RecordFile:
    """Fixed-size records, record index = iteration."""

    def __init__(self, path, dtype, shape):
        self.path, self.dtype, self.shape = path, np.dtype(dtype).newbyteorder('<'), tuple(int(s) for s in shape)
        self.nbytes = int(np.prod(self.shape, dtype=np.int64)) * self.dtype.itemsize

    def write(self, index, array):
        a = np.ascontiguousarray(array, dtype=self.dtype).reshape(self.shape)
        mode = 'r+b' if os.path.exists(self.path) else 'w+b'
        with open(self.path, mode) as fh:
            fh.seek(index * self.nbytes)
            fh.write(a.tobytes())

    def count(self):
        if not os.path.exists(self.path) or self.nbytes == 0:
            return 0
        return os.path.getsize(self.path) // self.nbytes

    def read(self, index=slice(None)):
        n = self.count()
        if self.nbytes == 0:
            data = np.zeros((n,) + self.shape, self.dtype)
        else:
            data = np.fromfile(self.path, dtype=self.dtype, count=n * int(np.prod(self.shape))).reshape((n,) + self.shape)
        return data[index]


class _RestrictedUnpickler(pickle.Unpickler):
    """The stored objects are this package's own state / move / System classes plus numpy arrays and builtin containers;
    nothing else may be constructed when a storage directory is opened (the reference stores YAML / XML, data only).

    ``find_class`` resolves ``name`` with getattr, and from protocol 4 on a dotted name walks attributes (``os.system`` reached
    through any module that imports os), so: no dotted names; from this package only classes DEFINED in this package; from
    numpy only the array / dtype / scalar constructors a pickled array needs (a whole-package whitelist admits gadgets such as
    numpy.testing._private.utils.runstring)."""
    _BUILTINS = {'dict', 'list', 'tuple', 'set', 'frozenset', 'int', 'float', 'complex', 'str', 'bytes', 'bytearray', 'bool',
                 'slice', 'range', 'object'}
    _NUMPY = {('numpy', 'ndarray'), ('numpy', 'dtype'),
              ('numpy.core.multiarray', '_reconstruct'), ('numpy._core.multiarray', '_reconstruct'),
              ('numpy.core.multiarray', 'scalar'), ('numpy._core.multiarray', 'scalar'),
              ('numpy.core.numeric', '_frombuffer'), ('numpy._core.numeric', '_frombuffer')}
    _NUMPY_SCALARS = {'bool_', 'int8', 'int16', 'int32', 'int64', 'uint8', 'uint16', 'uint32', 'uint64', 'float16', 'float32',
                      'float64', 'complex64', 'complex128', 'str_', 'bytes_'}
    _OTHER = {('collections', 'OrderedDict'), ('copyreg', '_reconstructor')}

    def find_class(self, module, name):
        refuse = pickle.UnpicklingError('storage refers to %s.%s, which is not a class this package stores' % (module, name))
        if '.' in name:
            raise refuse
        if module == 'builtins':
            if name in self._BUILTINS:
                return super().find_class(module, name)
            raise refuse
        if (module, name) in self._NUMPY or (module, name) in self._OTHER or (module == 'numpy' and name in self._NUMPY_SCALARS):
            return super().find_class(module, name)
        if module == 'openmmtools_amd' or module.startswith('openmmtools_amd.'):
            obj = super().find_class(module, name)
            if isinstance(obj, type) and (obj.__module__ == 'openmmtools_amd' or obj.__module__.startswith('openmmtools_amd.')):
                return obj
        raise refuse


def _reference_read(method):
    """Reading methods answer from the reference's own netCDF4 store when the reporter was opened on one (read-only)."""
    import functools

    @functools.wraps(method)
    def wrapper(self, *args, **kwargs):
        if self._ref is not None:
            return getattr(self._ref, method.__name__)(*args, **kwargs)
        return method(self, *args, **kwargs)
    return wrapper


class MultiStateReporter:
    """multistatereporter.py:69.  ``storage`` is a path; the analysis store is ``<storage>`` (a directory), the
    checkpoint store ``<storage stem>_checkpoint`` beside it (the reference: ``<name>.nc`` and
    ``<name>_checkpoint.nc``, :176-199)."""

    def __init__(self, storage, open_mode=None, checkpoint_interval=50, checkpoint_storage=None,
                 analysis_particle_indices=(), position_interval=1, velocity_interval=1, layout='auto'):
        self._storage_analysis = str(storage)
        stem = self._storage_analysis[:-3] if self._storage_analysis.endswith('.nc') else self._storage_analysis
        # multistatereporter.py:141-149: a checkpoint name is relative to the analysis file's directory (an absolute path stays)
        self._storage_checkpoint = (os.path.join(os.path.dirname(self._storage_analysis), str(checkpoint_storage)) if checkpoint_storage
                                    else stem + '_checkpoint')
        if type(checkpoint_interval) != int:                                   # multistatereporter.py:141-142
            raise ValueError("checkpoint_interval must be an integer!")
        self._checkpoint_interval = int(checkpoint_interval)
        self._analysis_particle_indices = tuple(int(i) for i in analysis_particle_indices)
        # multistatereporter.py:133-134, 1686-1692: how often the analysis file gets the flagged particles' positions / velocities
        # (0: never); checkpoints always carry both
        self._position_interval, self._velocity_interval = int(position_interval), int(velocity_interval)
        self._files = {}
        self._meta = None
        self._open_mode = None
        self._ref = None                  # a ReferenceStoreReader when `storage` is a netCDF4 file (the reference's layout)
        self._ncw = None                  # a ReferenceStoreWriter when this reporter WRITES that layout (storage path ends in .nc)
        # 'auto': the reference's netCDF4 layout for paths ending in .nc when libhdf5 is there, else the record container;
        # 'records' / 'netcdf4' force one (the sampler picks 'records' for states the netCDF4 layout cannot hold)
        if layout not in ('auto', 'records', 'netcdf4'):
            raise ValueError("layout must be 'auto', 'records' or 'netcdf4'")
        self.layout = layout
        if open_mode is not None:
            self.open(open_mode)

    # ---- life cycle (:240-330) -----------------------------------------------------------------------------
    @property
    def filepath(self):
        return self._storage_analysis

    @property
    def checkpoint_interval(self):
        return self._checkpoint_interval

    @property
    def analysis_particle_indices(self):
        """:214-225 and tests/test_sampling.py:816-866: what an open store already holds takes priority over the constructor's
        argument (the netCDF4 layout stores the indices; the record container keeps no per-particle trajectory)."""
        if self._ref is not None:
            stored = self._ref.analysis_particle_indices()
            if stored is not None:
                return stored
        return self._analysis_particle_indices

    @property
    def position_interval(self):
        """:228-230."""
        return self._position_interval

    @property
    def velocity_interval(self):
        """:233-235."""
        return self._velocity_interval

    def storage_exists(self, skip_size=False):
        """:237-261 (skip_size: the reference's guard against a zero-size netCDF file just created; the stores here are complete once
        they exist -- accepted, nothing to skip)."""
        from ._reference_store import is_reference_store
        return is_reference_store(self._storage_analysis) or os.path.exists(os.path.join(self._storage_analysis, 'meta.json'))

    @property
    def is_reference_store(self):
        return self._ref is not None

    def is_open(self):
        return self._open_mode is not None

    def open(self, mode='r', convention='ReplicaExchange', netcdf_format='NETCDF4'):
        """:280-340.  convention / netcdf_format: what the reference stamps on / asks of its netCDF file; the stores of the reference's
        layout written here carry exactly these two defaults, anything else is refused."""
        if convention != 'ReplicaExchange' or netcdf_format != 'NETCDF4':
            raise ValueError("only convention='ReplicaExchange' and netcdf_format='NETCDF4' (the reference's defaults) are written")
        if mode not in ('r', 'w', 'a'):
            raise ValueError("open mode must be 'r', 'w' or 'a'")
        from ._reference_store import is_reference_store, ReferenceStoreReader, ReferenceStoreWriter
        nc_path = self._storage_analysis.endswith('.nc') and self.layout != 'records' and not os.path.isdir(self._storage_analysis)
        if self.layout == 'netcdf4' and not self._storage_analysis.endswith('.nc'):
            raise ValueError("the netCDF4 layout needs a storage path ending in '.nc'")
        if nc_path and mode in ('w', 'a') and self.layout == 'auto':
            from . import _netcdf4_write
            if not _netcdf4_write.available():
                logger.warning('no libhdf5 / libhdf5_hl: %s is written as a record-file container, not as netCDF4', self._storage_analysis)
                nc_path = False
        if nc_path and mode in ('w', 'a') and (not is_reference_store(self._storage_analysis)
                                               or ReferenceStoreWriter.written_here(self._storage_analysis)):
            # the reference's own layout (netCDF4 = HDF5 through libhdf5): <name>.nc and <name>_checkpoint.nc, multistatereporter.py:176-199;
            # readable by the reference's reporter and analyzers.  Only files this package wrote are truncated or extended.
            stem = self._storage_analysis[:-3]
            ckpt = self._storage_checkpoint if self._storage_checkpoint.endswith('.nc') else stem + '_checkpoint.nc'
            for d in (os.path.dirname(os.path.abspath(self._storage_analysis)), os.path.dirname(os.path.abspath(ckpt))):
                os.makedirs(d, exist_ok=True)
            self._ncw = ReferenceStoreWriter(self._storage_analysis, ckpt, mode, self._checkpoint_interval,
                                             analysis_particle_indices=self._analysis_particle_indices,
                                             position_interval=self._position_interval, velocity_interval=self._velocity_interval)
            self._ref = self._ncw.reader()
            self._checkpoint_interval = self._ref.checkpoint_interval
            self._open_mode = mode
            self._meta = dict(format='netcdf4')
            return
        if is_reference_store(self._storage_analysis):
            # a store written by the reference itself (netCDF4): readable through libhdf5, never written (multistatereporter.py
            # of the reference owns that file); a simulation resumed from it reports into a store of its own
            if mode == 'w':
                raise IOError('{} is a netCDF4 store written by the reference: it can be read and resumed from, not overwritten'.format(self._storage_analysis))
            ckpt = self._storage_checkpoint if os.path.isfile(self._storage_checkpoint) else None
            self._ref = ReferenceStoreReader(self._storage_analysis, ckpt)
            self._checkpoint_interval = self._ref.checkpoint_interval
            self._open_mode = 'r'
            return
        if mode == 'r' and not self.storage_exists():
            raise OSError(f"{self._storage_analysis} does not exist")                 # :419
        if mode == 'w':
            # start a fresh store: remove only the files THIS format owns (never the contents of an unrelated directory)
            for d in (self._storage_analysis, self._storage_checkpoint):
                if os.path.isdir(d):
                    for f in os.listdir(d):
                        if self._owns(f):
                            os.remove(os.path.join(d, f))
        if mode in ('w', 'a'):
            os.makedirs(self._storage_analysis, exist_ok=True)
            os.makedirs(self._storage_checkpoint, exist_ok=True)
        self._open_mode = mode
        self._meta = None
        if self.storage_exists():
            with open(os.path.join(self._storage_analysis, 'meta.json')) as fh:
                self._meta = json.load(fh)
            self._checkpoint_interval = int(self._meta.get('checkpoint_interval', self._checkpoint_interval))
            self._declare()

    _OWNED_NAMES = re.compile(
        r'^(meta\.json|last_iteration\.json(\.tmp)?|(thermodynamic_states|mcmc_moves|options|metadata)\.pkl|online_\w+_\d{9}\.pkl|'
        r'(energies|unsampled_energies|timestamp|logZ|log_weights|f_k|f_k_offline|free_energy)\.f8|neighborhoods\.i1|'
        r'(states|accepted|proposed)\.i4|iteration_\d{9}(\.tmp)?\.npz)$')

    @classmethod
    def _owns(cls, filename):
        """Exactly the file names this container writes (typed record files, the four pickled objects, per-checkpoint online data,
        meta / last-iteration JSON, checkpoint npz) -- not every *.json / *.pkl / *.npz a user may keep in the same directory."""
        return cls._OWNED_NAMES.match(filename) is not None

    def close(self):
        if self._ncw is not None:
            self._ncw.close()
            self._ncw = self._ref = None
        elif self._ref is not None:
            self._ref.close()
            self._ref = None
        self._open_mode = None

    def sync(self):
        if self._ncw is not None:
            self._ncw.sync()                   # (the record files are flushed when each write closes its file)

    def _require_write(self):
        if self._ref is not None and self._ncw is None:
            raise IOError('a store written by the reference is read-only here')
        if self._open_mode not in ('w', 'a'):
            raise IOError('storage is not open for writing')

    def _declare(self):
        m = self._meta
        R, K, U = m['n_replicas'], m['n_states'], m['n_unsampled']
        d = self._storage_analysis
        self._files = {
            'energies': _RecordFile(os.path.join(d, 'energies.f8'), 'f8', (R, K)),                # :889-892
            'neighborhoods': _RecordFile(os.path.join(d, 'neighborhoods.i1'), 'i1', (R, K)),      # :893-897
            'unsampled_energies': _RecordFile(os.path.join(d, 'unsampled_energies.f8'), 'f8', (R, U)),
            'states': _RecordFile(os.path.join(d, 'states.i4'), 'i4', (R,)),                      # :806
            'accepted': _RecordFile(os.path.join(d, 'accepted.i4'), 'i4', (K, K)),                # :982-989
            'proposed': _RecordFile(os.path.join(d, 'proposed.i4'), 'i4', (K, K)),
            'timestamp': _RecordFile(os.path.join(d, 'timestamp.f8'), 'f8', ()),
            'logZ': _RecordFile(os.path.join(d, 'logZ.f8'), 'f8', (K,)),                          # online data, SAMS
            'log_weights': _RecordFile(os.path.join(d, 'log_weights.f8'), 'f8', (K,)),
            # online / offline free energy estimates (multistatesampler.py:1602-1605, 1662-1664)
            'f_k': _RecordFile(os.path.join(d, 'f_k.f8'), 'f8', (K,)),
            'f_k_offline': _RecordFile(os.path.join(d, 'f_k_offline.f8'), 'f8', (K + U,)),
            'free_energy': _RecordFile(os.path.join(d, 'free_energy.f8'), 'f8', (2,)),
        }

    def initialize(self, n_replicas, n_states, n_unsampled, n_atoms):
        """Dimensions of the record variables (the reference creates them lazily on first write)."""
        self._require_write()
        if self._ncw is not None:
            return                             # (netCDF variables are created with their first record, like the reference's)
        self._meta = dict(n_replicas=int(n_replicas), n_states=int(n_states), n_unsampled=int(n_unsampled),
                          n_atoms=int(n_atoms), checkpoint_interval=self._checkpoint_interval,
                          format='openmmtools_amd-records-1', created=time.time())
        with open(os.path.join(self._storage_analysis, 'meta.json'), 'w') as fh:
            json.dump(self._meta, fh)
        self._declare()

    # ---- states, moves, options, metadata (:476-690; serialised with pickle instead of YAML/XML) -------------
    def _write_object(self, name, obj):
        self._require_write()
        with open(os.path.join(self._storage_analysis, name + '.pkl'), 'wb') as fh:
            pickle.dump(obj, fh)

    def _read_object(self, name):
        with open(os.path.join(self._storage_analysis, name + '.pkl'), 'rb') as fh:
            return _RestrictedUnpickler(fh).load()

    def write_thermodynamic_states(self, thermodynamic_states, unsampled_states):
        if self._ncw is not None:
            self._require_write()
            return self._ncw.write_thermodynamic_states(thermodynamic_states, unsampled_states)
        self._write_object('thermodynamic_states', (list(thermodynamic_states), list(unsampled_states)))

    @_reference_read
    def read_thermodynamic_states(self):
        return self._read_object('thermodynamic_states')

    def write_mcmc_moves(self, mcmc_moves):
        if self._ncw is not None:
            self._require_write()
            return self._ncw.write_mcmc_moves(mcmc_moves)
        self._write_object('mcmc_moves', list(mcmc_moves))

    @_reference_read
    def read_mcmc_moves(self):
        return self._read_object('mcmc_moves')

    def write_dict(self, path, data, nested=False, fixed_dimension=False):
        """:1094-1115, 1817-1880 (``nested`` / ``fixed_dimension`` choose among the netCDF4 layout's three representations; the
        record container has one)."""
        if self._ncw is not None:
            self._require_write()
            return self._ncw.write_dict(path, data, nested=nested, fixed_dimension=fixed_dimension)
        self._write_object(path, dict(data))

    _write_dict = write_dict                      # the reference's tests call the private name

    @_reference_read
    def read_dict(self, path):
        """:1117-1165: a stored dictionary, or with 'name/key/subkey' one entry of it."""
        head, *keys = path.strip('/').split('/')
        value = self._read_object(head)
        for k in keys:
            value = value[k]
        return value

    # ---- per-iteration analysis data -----------------------------------------------------------------------
    def write_energies(self, energy_thermodynamic_states, energy_neighborhoods, energy_unsampled_states, iteration):
        """:865-929."""
        if self._ncw is not None:
            self._require_write()
            return self._ncw.write_energies(energy_thermodynamic_states, energy_neighborhoods, energy_unsampled_states, iteration)
        self._require_write()
        self._files['energies'].write(iteration, energy_thermodynamic_states)
        self._files['neighborhoods'].write(iteration, energy_neighborhoods)
        self._files['unsampled_energies'].write(iteration, energy_unsampled_states)

    def _calculate_checkpoint_iteration(self, iteration):
        """:1504-1515: the frame of ``iteration`` in the checkpoint store, None off the checkpoint interval."""
        index, remainder = divmod(int(iteration), self._checkpoint_interval)
        return int(index) if remainder == 0 else None

    def _map_iteration_to_good(self, iteration):
        """:1517-1541: an index or slice over the iterations that were written COMPLETELY (0 .. last_iteration): negative
        indices count back from the last good iteration, slices stop there, an index beyond it raises IndexError -- records a
        crashed or abandoned continuation left behind the last good iteration are never served."""
        last_good = self.read_last_iteration(last_checkpoint=False)
        return np.arange((last_good if last_good is not None else -1) + 1, dtype=int)[iteration]

    @_reference_read
    def read_energies(self, iteration=slice(None)):
        """:817-863 -> (energy_thermodynamic_states, neighborhoods, energy_unsampled_states)."""
        iteration = self._map_iteration_to_good(iteration)
        e = self._files['energies'].read(iteration)
        fu = self._files['unsampled_energies']
        eu = fu.read(iteration) if fu.nbytes else np.zeros(e.shape[:-1] + (0,), fu.dtype)    # no unsampled states: empty records
        return e, self._files['neighborhoods'].read(iteration), eu

    def write_replica_thermodynamic_states(self, state_indices, iteration):
        """:797-815."""
        if self._ncw is not None:
            self._require_write()
            return self._ncw.write_replica_thermodynamic_states(state_indices, iteration)
        self._require_write()
        self._files['states'].write(iteration, state_indices)

    @_reference_read
    def read_replica_thermodynamic_states(self, iteration=slice(None)):
        """:775-795."""
        return self._files['states'].read(self._map_iteration_to_good(iteration)).astype(np.int64)

    def write_mixing_statistics(self, n_accepted_matrix, n_proposed_matrix, iteration):
        """:957-999 (stored as i4, like the reference)."""
        if self._ncw is not None:
            self._require_write()
            return self._ncw.write_mixing_statistics(n_accepted_matrix, n_proposed_matrix, iteration)
        self._require_write()
        self._files['accepted'].write(iteration, n_accepted_matrix)
        self._files['proposed'].write(iteration, n_proposed_matrix)

    @_reference_read
    def read_mixing_statistics(self, iteration=slice(None)):
        """:931-955."""
        iteration = self._map_iteration_to_good(iteration)
        return self._files['accepted'].read(iteration), self._files['proposed'].read(iteration)

    def write_timestamp(self, iteration):
        """:1001-1018."""
        if self._ncw is not None:
            self._require_write()
            return self._ncw.write_timestamp(iteration)
        self._require_write()
        self._files['timestamp'].write(iteration, time.time())

    @_reference_read
    def read_timestamp(self, iteration=slice(None)):
        return self._files['timestamp'].read(self._map_iteration_to_good(iteration))

    def write_online_data_dynamic_and_static(self, iteration, **kwargs):
        """:1167-1252 (the variables SAMS writes: logZ, log_weights)."""
        if self._ncw is not None:
            self._require_write()
            return self._ncw.write_online_analysis(iteration, **kwargs)
        self._require_write()
        for k, v in kwargs.items():
            if k in self._files and v is not None:
                self._files[k].write(iteration, v)
            elif v is not None and iteration % self._checkpoint_interval == 0:
                self._write_object('online_%s_%09d' % (k, iteration), v)     # small python objects: checkpoint iterations only

    def write_online_analysis_data(self, iteration, **kwargs):
        """:1305-1339: 1-D numeric online-analysis variables, per iteration (``iteration`` None: the static copy, which
        here is simply the latest record)."""
        if iteration is None:
            return
        self.write_online_data_dynamic_and_static(iteration, **kwargs)

    @_reference_read
    def read_online_analysis_data(self, iteration, *keys):
        """:1236-1303: {key: value} at ``iteration`` (None: the most recent record).  KeyError for an unknown variable,
        IndexError when nothing was written at that iteration — the two exceptions the sampler's reader handles."""
        out = {}
        for k in keys:
            if k not in self._files:
                raise KeyError(k)
            n = self._files[k].count()
            if n == 0:
                raise ValueError('no {} in storage'.format(k))
            idx = n - 1 if iteration is None else int(iteration)
            if idx >= n or idx < 0:
                raise IndexError(idx)
            out[k] = self._files[k].read(idx)
        return out

    # ---- small accessors of the reference built on the readers above (they answer from whichever store is open) ----------------
    @property
    def n_states(self):
        """:197-201: number of sampled thermodynamic states (None while closed)."""
        if not self.is_open():
            return None
        return len(self.read_thermodynamic_states()[0])

    @property
    def n_replicas(self):
        """:203-207."""
        if not self.is_open():
            return None
        return int(np.asarray(self.read_replica_thermodynamic_states(iteration=0)).shape[-1])

    @property
    def is_periodic(self):
        """:209-215: whether the stored configurations carry box vectors."""
        if not self.is_open():
            return None
        return self.read_sampler_states(0)[0].box_vectors is not None

    def read_end_thermodynamic_states(self):
        """:480-560: the unsampled states if there are any, else the first and the last sampled state."""
        states_, unsampled = self.read_thermodynamic_states()
        return list(unsampled) if len(unsampled) > 0 else [states_[0], states_[-1]]

    def read_logZ(self, iteration):
        """:1203-1220 (SAMS)."""
        return self.read_online_analysis_data(iteration, 'logZ')['logZ']

    def write_logZ(self, iteration, logZ):
        """:1222-1234."""
        self.write_online_data_dynamic_and_static(iteration, logZ=logZ)

    def write_current_statistics(self, data):
        """:1353-1375: appends one YAML document per call to ``<storage stem>_real_time_analysis.yaml``."""
        self._require_write()
        import yaml
        stem = self._storage_analysis[:-3] if self._storage_analysis.endswith('.nc') else self._storage_analysis
        with open(stem + '_real_time_analysis.yaml', 'a') as fh:
            fh.write(yaml.dump([data], sort_keys=False))

    @_reference_read
    def read_online_data_if_present(self, iteration):
        out = {}
        for k in ('logZ', 'log_weights'):
            if self._files[k].count() > iteration:
                out[k] = self._files[k].read(iteration)
        suffix = '_%09d.pkl' % iteration
        for f in os.listdir(self._storage_analysis):
            if f.startswith('online_') and f.endswith(suffix):
                out[f[len('online_'):-len(suffix)]] = self._read_object(f[:-4])
        return out or None

    def write_last_iteration(self, iteration):
        """:1072-1092: marks the last iteration all of whose data is on disk."""
        if self._ncw is not None:
            self._require_write()
            return self._ncw.write_last_iteration(iteration)
        self._require_write()
        tmp = os.path.join(self._storage_analysis, 'last_iteration.json.tmp')
        with open(tmp, 'w') as fh:
            json.dump(dict(last_iteration=int(iteration)), fh)
        os.replace(tmp, os.path.join(self._storage_analysis, 'last_iteration.json'))

    @_reference_read
    def read_last_iteration(self, last_checkpoint=True):
        """:1020-1070: the last good iteration, or (default) the last one with a checkpoint at or below it."""
        path = os.path.join(self._storage_analysis, 'last_iteration.json')
        if not os.path.exists(path):
            return None
        with open(path) as fh:
            last = int(json.load(fh)['last_iteration'])
        if not last_checkpoint:
            return last
        cps = [c for c in self.read_checkpoint_iterations() if c <= last]
        return max(cps) if cps else None

    # ---- checkpoints (:1597-1737) --------------------------------------------------------------------------
    def _checkpoint_path(self, iteration):
        return os.path.join(self._storage_checkpoint, 'iteration_%09d.npz' % iteration)

    @_reference_read
    def read_checkpoint_iterations(self):
        if not os.path.isdir(self._storage_checkpoint):
            return []
        return sorted(int(f[10:19]) for f in os.listdir(self._storage_checkpoint) if f.startswith('iteration_') and f.endswith('.npz'))

    def write_sampler_states(self, sampler_states, iteration):
        """:1094-1115: only on checkpoint iterations; positions / velocities as f4 (:1621-1632), box f4."""
        if self._ncw is not None:
            self._require_write()
            return self._ncw.write_sampler_states(sampler_states, iteration)
        self._require_write()
        if iteration % self._checkpoint_interval != 0:
            return False
        x = np.stack([np.asarray(s.positions, dtype=np.float64) for s in sampler_states]).astype('<f4')
        v = np.stack([np.zeros_like(np.asarray(s.positions, dtype=np.float64)) if s.velocities is None
                      else np.asarray(s.velocities, dtype=np.float64) for s in sampler_states]).astype('<f4')
        has_box = all(s.box_vectors is not None for s in sampler_states)
        box = np.stack([np.asarray(s.box_vectors, dtype=np.float64) for s in sampler_states]).astype('<f4') if has_box \
            else np.zeros((len(sampler_states), 3, 3), '<f4')
        tmp = self._checkpoint_path(iteration) + '.tmp.npz'
        np.savez(tmp, positions=x, velocities=v, box_vectors=box, has_box=np.array(has_box))
        os.replace(tmp, self._checkpoint_path(iteration))
        return True

    @_reference_read
    def read_sampler_states(self, iteration, analysis_particles_only=False):
        """:1117-1165: None when ``iteration`` is not a checkpoint iteration."""
        path = self._checkpoint_path(iteration)
        if not os.path.exists(path):
            return None
        from .. import states
        with np.load(path) as d:
            x, v, box, has_box = d['positions'], d['velocities'], d['box_vectors'], bool(d['has_box'])
        return [states.SamplerState(x[r].astype(np.float64), velocities=v[r].astype(np.float64),
                                    box_vectors=box[r].astype(np.float64) if has_box else None) for r in range(len(x))]

    # ---- the sampler's hook: what MultiStateSampler._report_iteration does (multistatesampler.py:1189-1222) --
    def write_iteration(self, sampler):
        # positions are needed on checkpoint iterations -- and at every iteration when particles are flagged for the analysis file
        # (multistatereporter.py:722-741; the netCDF4 layout only)
        every = bool(self._analysis_particle_indices) and str(self._storage_analysis).endswith('.nc') and self.layout != 'records'
        if getattr(sampler, '_comm', None) is not None and sampler._comm.rank != 0:
            if every or sampler._iteration % self._checkpoint_interval == 0:
                sampler._gather_sampler_states()          # collective: every rank takes part
            return
        it = sampler._iteration
        if self._meta is None:
            self.initialize(sampler.n_replicas, sampler.n_states, len(sampler._unsampled_states),
                            sampler._thermodynamic_states[0].n_particles)
        if every or it % self._checkpoint_interval == 0:
            sampler._gather_sampler_states()
            self.write_sampler_states(sampler._sampler_states, it)                                  # :1217
        self.write_replica_thermodynamic_states(sampler._replica_thermodynamic_states, it)          # :1218
        self.write_mcmc_moves(sampler._mcmc_moves)                                                  # :1219
        self.write_energies(sampler._energy_thermodynamic_states, sampler._neighborhoods,
                            sampler._energy_unsampled_states, it)                                   # :1220
        self.write_mixing_statistics(sampler._n_accepted_matrix, sampler._n_proposed_matrix, it)    # :1221
        online = sampler._online_data() if hasattr(sampler, '_online_data') else None
        if online:
            self.write_online_data_dynamic_and_static(it, **online)
        self.write_timestamp(it)                                                                    # :1202
        self.write_last_iteration(it)                                                               # :1207

"""Read-only access to a store WRITTEN BY THE REFERENCE (openmmtools.multistate.MultiStateReporter, netCDF4 = HDF5), with the
reading interface of this package's MultiStateReporter, so that ``from_storage`` can resume a reference simulation on the
device and the analyzer can read its energies.

Layout read (multistatereporter.py of the reference):
    analysis file  <name>.nc            :280-460  variables energies f8[iteration, replica, state] :865-930,
                                        neighborhoods i1 :893-901, unsampled_energies :911-925, states i4[iteration, replica]
                                        :763-793, accepted / proposed i4[iteration, state, state] :957-999, timestamp, last_iteration
                                        :1184-1201, options / metadata / mcmc_moves/move<k> / thermodynamic_states/state<k> /
                                        unsampled_states/state<k> as YAML dictionaries :1817-1880 (character arrays or strings),
                                        global attribute CheckpointInterval
    checkpoint     <name>_checkpoint.nc :1597-1737 positions / velocities f4[iteration, replica, atom, 3], box_vectors
                                        f4[iteration, replica, 3, 3], volumes; legacy files (< 0.21.3) carry no velocities
                                        and are read as zero velocities :1795-1806 (tests/test_sampling.py:2943-2990)
Serialized objects (utils.serialize / deserialize): '_serialized__class_name', '_serialized__module_name' + the object's
__getstate__; quantities as '!Quantity {unit, value}'; a ThermodynamicState's System as zlib-compressed OpenMM XML under
'standard_system', shared between compatible states through '_Reporter__compatible_state' :552-610.

What is understood: ThermodynamicState (temperature, pressure) on a System of the forces openmmtools_amd/system_xml.py reads,
CompoundThermodynamicState with one AlchemicalState (round 4: the System as the alchemical factory builds it, _alchemical_xml.py),
LangevinSplittingDynamicsMove / LangevinDynamicsMove, MultiStateSampler / ReplicaExchangeSampler / ParallelTemperingSampler /
SAMSSampler options.  Anything else (other composable states, other moves) raises NotImplementedError naming it.  The HDF5
library is loaded through ctypes (``_hdf5``); without it the store cannot be opened (ImportError says so).
"""
import os
import zlib

import numpy as np

from . import _hdf5

_UNIT_TO_MD = {            # factor to the MD unit system (nm, ps, amu, kJ/mol, K, bar)
    'kelvin': 1.0, 'nanometer': 1.0, 'picosecond': 1.0, '/picosecond': 1.0, 'femtosecond': 1e-3, 'angstrom': 0.1,
    'bar': 0.0602214076, 'atmosphere': 0.0602214076 * 1.01325,     # kJ/mol/nm^3 (pressure x N_A), the engine's pressure unit
    'dimensionless': 1.0, 'nanometer/picosecond': 1.0, 'kilojoule/mole': 1.0,
    'kilocalorie/mole': 4.184, '/femtosecond': 1e3,
}


def _yaml_load(text):
    import yaml

    class Loader(yaml.SafeLoader):
        pass

    def quantity(loader, node):
        d = loader.construct_mapping(node, deep=True)
        unit_name = str(d['unit']).replace(' ', '')
        if unit_name not in _UNIT_TO_MD:
            raise NotImplementedError('unit %r in a reference store' % d['unit'])
        v = d['value']
        return (np.asarray(v, dtype=np.float64) if isinstance(v, (list, tuple)) else float(v)) * _UNIT_TO_MD[unit_name]

    def ndarray(loader, node):
        d = loader.construct_mapping(node, deep=True)
        return np.asarray(d.get('values', d.get('value')), dtype=d.get('type', None))

    Loader.add_constructor('!Quantity', quantity)
    Loader.add_constructor('!ndarray', ndarray)
    return yaml.load(text, Loader=Loader)


def is_reference_store(path):
    """A file (not this package's record directory) that starts with the HDF5 signature."""
    path = str(path)
    if not os.path.isfile(path):
        return False
    with open(path, 'rb') as fh:
        return fh.read(8) == b'\x89HDF\r\n\x1a\n'


class ReferenceStoreReader:
    """The read half of MultiStateReporter for a reference netCDF4 store.  Iteration arguments follow the reference:
    an int or a slice over the iterations up to the last good one."""

    def __init__(self, analysis_path, checkpoint_path=None):
        self._path = str(analysis_path)
        stem = self._path[:-3] if self._path.endswith('.nc') else self._path
        self._cpath = str(checkpoint_path) if checkpoint_path else stem + '_checkpoint.nc'
        self._a = _hdf5.File(self._path)
        self._c = _hdf5.File(self._cpath) if os.path.isfile(self._cpath) else None
        self._check_uuid()
        ci = self._a.attr('CheckpointInterval')
        self._checkpoint_interval = int(np.asarray(ci).reshape(-1)[0]) if ci is not None else 1

    def _check_uuid(self):
        """multistatereporter.py:318-354: the checkpoint file must carry the analysis file's UUID (a checkpoint of another
        simulation next to this analysis file is refused)."""
        if self._c is None:
            return
        ua, uc = self._a.attr('UUID'), self._c.attr('UUID')
        if ua is not None and uc is not None and str(np.asarray(ua).reshape(-1)[0]) != str(np.asarray(uc).reshape(-1)[0]):
            a, c = str(np.asarray(ua).reshape(-1)[0]), str(np.asarray(uc).reshape(-1)[0])
            self.close()
            raise OSError('Checkpoint UUID does not match analysis UUID! This checkpoint file came from another simulation!\n'
                          'Analysis UUID: {}; Checkpoint UUID: {}'.format(a, c))

    filepath = property(lambda self: self._path)
    title = property(lambda self: self._a.attr('title'))
    checkpoint_interval = property(lambda self: self._checkpoint_interval)

    def analysis_particle_indices(self):
        """The stored indices (an empty tuple when none were flagged), None for a file that predates the variable."""
        if '/analysis_particle_indices' not in self._a:
            return None
        return tuple(int(i) for i in np.asarray(self._a.read('/analysis_particle_indices')).reshape(-1))

    def read_seed(self):
        """The Philox seed of a store this package wrote (global attribute; None for the reference's own files)."""
        v = self._a.attr('openmmtools_amd_seed')
        return None if v is None else int(np.asarray(v).reshape(-1)[0])

    def close(self):
        self._a.close()
        if self._c is not None:
            self._c.close()

    # ---- dictionaries ------------------------------------------------------------------------------------------
    def read_dict(self, path):
        """multistatereporter.py:1043-1112: a variable holding YAML (character array or string), or a group of them."""
        path = '/' + path.strip('/')
        if self._a.is_group(path):
            groups, datasets = self._a.keys(path)
            data = {k: self.read_dict(path + '/' + k) for k in groups + datasets}
            if path == '/metadata':
                data['title'] = self.title
            return data
        if path not in self._a:
            head, key = path.rsplit('/', 1)
            if not head:
                raise KeyError(path)
            return self.read_dict(head)[key]
        raw = self._a.read(path)
        text = raw.tobytes().decode() if getattr(raw, 'dtype', None) is not None and raw.dtype.kind == 'S' else str(np.asarray(raw).reshape(-1)[0])
        data = _yaml_load(text)
        if path == '/metadata' and isinstance(data, dict):
            data['title'] = self.title
        return data

    # ---- iterations --------------------------------------------------------------------------------------------
    def read_last_iteration(self, last_checkpoint=True):
        last = int(np.asarray(self._a.read('/last_iteration')).reshape(-1)[0])
        if last_checkpoint:                                             # :1172-1180
            for i in range(last, -1, -1):
                if i % self._checkpoint_interval == 0:
                    return i
        return last

    def read_checkpoint_iterations(self):
        last = self.read_last_iteration(last_checkpoint=False)
        return [i for i in range(0, last + 1) if i % self._checkpoint_interval == 0]

    def _clip(self, array, iteration):
        last = self.read_last_iteration(last_checkpoint=False)
        return array[:last + 1][iteration]

    def read_energies(self, iteration=slice(None)):
        e = self._clip(self._a.read('/energies'), iteration)
        nb = self._clip(self._a.read('/neighborhoods'), iteration) if '/neighborhoods' in self._a else np.ones(e.shape, 'i1')
        if '/unsampled_energies' in self._a:
            eu = self._clip(self._a.read('/unsampled_energies'), iteration)
        else:
            eu = np.zeros(e.shape[:-1] + (0,))
        return e, nb, eu

    def read_replica_thermodynamic_states(self, iteration=slice(None)):
        return self._clip(self._a.read('/states'), iteration).astype(np.int64)

    def read_mixing_statistics(self, iteration=slice(None)):
        return self._clip(self._a.read('/accepted'), iteration), self._clip(self._a.read('/proposed'), iteration)

    def read_timestamp(self, iteration=slice(None)):
        return self._clip(self._a.read('/timestamp'), iteration)

    def read_online_data_if_present(self, iteration):
        out = {}
        for key in ('logZ', 'log_weights', 'stage', 't0', 'state_histogram'):
            for path in ('/online_analysis/' + key + '_history', '/online_analysis/' + key):
                if path in self._a:
                    a = self._a.read(path)
                    out[key] = a[iteration] if a.ndim == 2 and a.shape[0] > iteration else a
                    break
        if 'stage' in out and 't0' in out:                          # sams.py:374-393: what a SAMS sampler restores
            out['sams_state'] = dict(stage=int(np.asarray(out.pop('stage')).reshape(-1)[0]), t0=int(np.asarray(out.pop('t0')).reshape(-1)[0]),
                                     histogram=None if 'state_histogram' not in out else np.asarray(out.pop('state_histogram')).astype(np.int64),
                                     iteration=int(iteration))
        return out or None

    def read_online_analysis_data(self, iteration, *keys):
        out = {}
        for key in keys:
            hist, stat = '/online_analysis/%s_history' % key, '/online_analysis/%s' % key
            if hist in self._a and self._a.shape(hist)[0] > iteration:
                out[key] = self._a.read(hist)[iteration]
            elif stat in self._a:
                out[key] = self._a.read(stat)
            else:
                raise KeyError(key)
        return out

    # ---- objects -----------------------------------------------------------------------------------------------
    def read_thermodynamic_states(self):
        """:552-610.  Returns (thermodynamic_states, unsampled_states) as this package's ThermodynamicState objects; states that
        name a compatible earlier state share its System object."""
        from .. import states, system_xml
        out, systems = {'thermodynamic_states': [], 'unsampled_states': []}, {}
        for kind in out:
            if not self._a.is_group('/' + kind):
                continue
            n = len(self._a.keys('/' + kind)[1])
            for k in range(n):
                outer = self.read_dict('%s/state%d' % (kind, k))
                d = outer
                while 'thermodynamic_state' in d:                    # CompoundThermodynamicState.__getstate__, states.py:2956-2971
                    d = d['thermodynamic_state']
                if d.get('_serialized__class_name') != 'ThermodynamicState':
                    raise NotImplementedError('state class %r in a reference store' % d.get('_serialized__class_name'))
                ref = d.get('_Reporter__compatible_state')
                if ref is None:
                    xml = zlib.decompress(d['standard_system']).decode()
                    system, _ = system_xml.from_xml(xml)
                    systems['%s/%d' % (kind, k)] = system
                else:
                    system = systems[ref]
                state = states.ThermodynamicState(system, float(d['temperature']),
                                                  pressure=None if d.get('pressure') is None else float(d['pressure']))
                if outer is not d:
                    composable = outer.get('composable_states', [])
                    names = [str(c.get('_serialized__class_name')) for c in composable]
                    if not names or any(n != 'AlchemicalState' for n in names):
                        raise NotImplementedError('compound thermodynamic states (%s) in a reference store' % ', '.join(names))
                    alchs = []
                    for c in composable:
                        if c.get('function_variables'):
                            raise NotImplementedError('AlchemicalState with alchemical functions')
                        par = {n: v for n, v in c['parameters'].items() if v is not None}     # None: not defined on the System (alchemy.py:94-99)
                        if any(isinstance(v, str) for v in par.values()):
                            raise NotImplementedError('AlchemicalState parameters given as functions')
                        alchs.append(states.AlchemicalState(parameters_name_suffix=c.get('parameters_name_suffix'), **par))
                    state = states.CompoundThermodynamicState(state, alchs)
                out[kind].append(state)
        return out['thermodynamic_states'], out['unsampled_states']

    def read_mcmc_moves(self):
        """:795-811."""
        from .. import mcmc
        n = len(self._a.keys('/mcmc_moves')[1])
        moves = []
        for k in range(n):
            d = self.read_dict('mcmc_moves/move%d' % k)
            name = d.get('_serialized__class_name')
            if name not in ('LangevinSplittingDynamicsMove', 'LangevinDynamicsMove'):
                raise NotImplementedError('MCMC move %r in a reference store' % name)
            kw = dict(timestep=float(d['timestep']), collision_rate=float(d['collision_rate']), n_steps=int(d['n_steps']),
                      reassign_velocities=bool(d['reassign_velocities']),
                      constraint_tolerance=float(d.get('constraint_tolerance', 1e-8)))
            if name == 'LangevinSplittingDynamicsMove':
                kw['splitting'] = str(d.get('splitting', 'V R O R V'))
                move = mcmc.LangevinSplittingDynamicsMove(**kw)
            else:
                move = mcmc.LangevinDynamicsMove(**kw)
            move.n_restart_attempts = int(d.get('n_restart_attempts', 4))
            moves.append(move)
        return moves

    def read_sampler_states(self, iteration, analysis_particles_only=False):
        """:670-705, 1739-1815: the checkpoint frame of ``iteration`` (None off the checkpoint interval); zero velocities where
        the file carries none."""
        from .. import states
        if analysis_particles_only:
            # :696-705: the subset the analysis file carries for every iteration
            if '/positions' not in self._a:
                raise ValueError('No particles were flagged for special analysis! No such trajectory would have been written!')
            def frame(name, like=None):
                # frames an interval skipped (or that lie behind the last one written) read as zeros, like netCDF's masked fill
                data = self._a.read(name) if name in self._a else None
                if data is None or iteration >= data.shape[0]:
                    return np.zeros(like.shape if like is not None else (0, 0, 3))
                out = np.array(data[iteration], dtype=np.float64)
                out[~np.isfinite(out) | (np.abs(out) > 1e30)] = 0.0
                return out
            x = frame('/positions')
            v = frame('/velocities', like=x)
            if x.size == 0 and v.size:
                x = np.zeros_like(v)
            box = None
            if '/box_vectors' in self._a:
                data = self._a.read('/box_vectors')
                if iteration < data.shape[0] and np.abs(data[iteration]).max() > 0 and np.abs(data[iteration]).max() < 1e30:
                    box = np.array(data[iteration], dtype=np.float64)
            return [states.SamplerState(x[r], velocities=v[r], box_vectors=None if box is None else box[r]) for r in range(x.shape[0])]
        if self._c is None:
            raise IOError('checkpoint file %s is missing' % self._cpath)
        if iteration % self._checkpoint_interval != 0:
            return None
        frame = iteration // self._checkpoint_interval
        pos = self._c.read('/positions')
        if frame >= pos.shape[0]:
            raise IndexError('no checkpoint frame for iteration %d' % iteration)
        x = pos[frame].astype(np.float64)
        v = self._c.read('/velocities')[frame].astype(np.float64) if '/velocities' in self._c else np.zeros_like(x)
        box = self._c.read('/box_vectors')[frame].astype(np.float64) if '/box_vectors' in self._c else None
        out = []
        for r in range(x.shape[0]):
            out.append(states.SamplerState(x[r], velocities=v[r], box_vectors=None if box is None else box[r]))
        return out


# =====================================================================================================================
# Writing the same layout (multistatereporter.py:280-460, 476-690, 865-1115, 1597-1737, 1817-1880)
# =====================================================================================================================

class _Quantity:
    """A number with the unit name the reference's YAML carries ('!Quantity {unit, value}')."""

    def __init__(self, value, unit):
        self.value, self.unit = value, unit


def _yaml_dump(data):
    import yaml

    class Dumper(yaml.SafeDumper):
        pass

    def quantity(dumper, q):
        v = q.value.tolist() if isinstance(q.value, np.ndarray) else float(q.value)
        return dumper.represent_mapping('!Quantity', {'unit': q.unit, 'value': v})

    def ndarray(dumper, a):
        # multistatereporter.py:1932-1939: dtype, shape and the values as nested lists under the !ndarray tag
        return dumper.represent_mapping('!ndarray', {'type': str(a.dtype), 'shape': list(a.shape), 'values': a.tolist()})

    Dumper.add_representer(_Quantity, quantity)
    Dumper.add_representer(np.ndarray, ndarray)
    Dumper.add_representer(np.float64, lambda d, x: d.represent_float(float(x)))
    Dumper.add_representer(np.int64, lambda d, x: d.represent_int(int(x)))
    Dumper.add_representer(np.bool_, lambda d, x: d.represent_bool(bool(x)))
    return yaml.dump(data, Dumper=Dumper)


PROGRAM = 'openmmtools_amd'
_STANDARD_TEMPERATURE = 273.0              # states.py:224-226: what the standard System's barostat carries
_STANDARD_PRESSURE_BAR = 1.0


class ReferenceStoreWriter:
    """The write half: the variables, dimensions, attributes and serialized objects the reference's MultiStateReporter
    creates through netCDF4-python, created through libhdf5 (``_netcdf4_write``).  What can be written is what
    ``ReferenceStoreReader`` understands: plain ThermodynamicStates (temperature, pressure) on Systems of the hot-path
    forces, Langevin moves, the samplers' options; anything else raises NotImplementedError naming it."""

    def __init__(self, analysis_path, checkpoint_path, mode, checkpoint_interval, title=None, analysis_particle_indices=(),
                 position_interval=1, velocity_interval=1):
        from . import _netcdf4_write as nw
        self._nw = nw
        self._path, self._cpath = str(analysis_path), str(checkpoint_path)
        self._interval = int(checkpoint_interval)
        self._position_interval, self._velocity_interval = int(position_interval), int(velocity_interval)
        fresh = mode == 'w' or not os.path.isfile(self._path)
        self._a = nw.NetCDF4File(self._path, 'w' if fresh else 'a')
        self._c = nw.NetCDF4File(self._cpath, 'w' if fresh or not os.path.isfile(self._cpath) else 'a')
        if not fresh:
            self.reader()._check_uuid()            # multistatereporter.py:343-354: a checkpoint of another simulation is refused
        if fresh:
            import time
            import uuid
            uid = str(uuid.uuid4())
            title = title or 'Multi-state sampler simulation created using %s on %s' % (PROGRAM, time.asctime())
            for f, used in ((self._a, 'analysis'), (self._c, 'checkpoint')):        # :432-460 _initialize_storage_file
                f.create_dimension('/scalar', 1)
                f.create_dimension('/iteration', None)
                f.create_dimension('/spatial', 3)
                f.set_attr('/', 'program', PROGRAM)
                f.set_attr('/', 'programVersion', '0.3')
                f.set_attr('/', 'Conventions', 'ReplicaExchange')
                f.set_attr('/', 'ConventionVersion', '0.2')
                f.set_attr('/', 'DataUsedFor', used)
                f.set_attr('/', 'CheckpointInterval', np.array([self._interval], dtype=np.int64))
                f.set_attr('/', 'UUID', uid)                                        # :346-361: the two files carry one UUID
                f.set_attr('/', 'title', title)
                f.set_attr('/', 'PositionInterval', np.array([self._position_interval], dtype=np.int64))       # :450-451
                f.set_attr('/', 'VelocityInterval', np.array([self._velocity_interval], dtype=np.int64))
                f.create_variable('/last_iteration', 'i8', ('scalar',))
                f.write('/last_iteration', [0])
            # :369-381: the reference's open() creates this variable when it is missing -- which fails on a file opened for
            # reading, so it has to be there from the start (no analysis particles = an unlimited dimension of length 0)
            idx = np.asarray(sorted(int(i) for i in analysis_particle_indices), dtype=np.int64)
            self._a.create_dimension('/analysis_particles', len(idx) if len(idx) else None)
            self._a.create_variable('/analysis_particle_indices', 'i8', ('analysis_particles',))
            self._a.set_attr('/analysis_particle_indices', 'long_name', 'analysis_particle_indices[analysis_particles] is the indices of the '
                             'particles with extra information stored about them in theanalysis file.')
            if len(idx):
                self._a.write('/analysis_particle_indices', idx)
            self._analysis_particles = tuple(idx.tolist())
        else:
            ci = self._a.attr('CheckpointInterval')
            if ci is not None:
                self._interval = int(np.asarray(ci).reshape(-1)[0])
            self._analysis_particles = tuple(int(i) for i in np.asarray(self._a.read('/analysis_particle_indices')).reshape(-1)) \
                if '/analysis_particle_indices' in self._a else ()

    @staticmethod
    def can_store(thermodynamic_states, unsampled_states, mcmc_moves):
        """None when this layout can hold the objects, else what it cannot (other state classes, other moves, alchemical Systems
        outside the factory's defaults)."""
        from .. import system_xml
        seen = set()
        for s in list(thermodynamic_states) + list(unsampled_states):
            if type(s).__name__ not in ('ThermodynamicState', 'CompoundThermodynamicState'):
                return type(s).__name__
            if (getattr(s.system, 'alchemical_region', None) is not None or getattr(s.system, 'alchemical_regions', None) is not None) and id(s.system) not in seen:
                seen.add(id(s.system))
                try:                                        # the factory's force set for this System (_alchemical_xml.py) or why not
                    system_xml.to_xml(s.system)
                except NotImplementedError as err:
                    return 'alchemical System: %s' % err
        for m in mcmc_moves:
            if type(m).__name__ not in ('LangevinSplittingDynamicsMove', 'LangevinDynamicsMove'):
                return type(m).__name__
        return None

    @staticmethod
    def written_here(path):
        """Only stores this writer made are extended in place; one written by the reference is read and continued elsewhere."""
        try:
            with _hdf5.File(str(path)) as f:
                return str(f.attr('program', default='')).startswith(PROGRAM)
        except Exception:
            return False

    def reader(self):
        r = ReferenceStoreReader.__new__(ReferenceStoreReader)
        r._path, r._cpath, r._a, r._c = self._path, self._cpath, self._a, self._c
        r._checkpoint_interval = self._interval
        return r

    def sync(self):
        self._a.flush()
        self._c.flush()

    def close(self):
        self._a.close()
        self._c.close()

    # ---- dictionaries (:1817-1880) -----------------------------------------------------------------------------
    def _write_text(self, f, path, text, fixed):
        group = path.rsplit('/', 1)[0]
        if group:
            f.create_group(group)
        if fixed:
            dim = '/fixedL%d' % len(text)
            if path not in f:
                f.create_dimension(dim, len(text))
                f.create_variable(path, 'S1', (dim[1:],))
            elif f.shape(path)[0] != len(text):
                raise IOError('%s: a fixed-length dictionary cannot change its length' % path)
            f.write(path, text)
        else:
            if path not in f:
                f.create_variable(path, 'str', ('scalar',))
            f.write(path, text)

    def write_dict(self, name, data, nested=False, fixed_dimension=False):
        """:1817-1880 _write_dict: one YAML string variable, or (``nested``) a group per dictionary and a variable per value;
        ``fixed_dimension`` stores the text as a character array of its own length instead of a variable-length string."""
        data = dict(data) if data is not None else {}
        if nested and name not in ('options', 'metadata'):
            self._write_nested('/' + name.strip('/'), data, fixed=bool(fixed_dimension))
            return
        if name == 'options' and 'kwargs' in data and 'cls' in data:
            # the reference restores with cls(**options) (multistatesampler.py:948-950): only its constructor's keywords may
            # appear; this package's extras (sampler class, Philox seed) go to global attributes
            self._a.set_attr('/', 'openmmtools_amd_sampler', '%s.%s' % (data.get('module', ''), data['cls']))
            if data.get('seed') is not None:
                self._a.set_attr('/', 'openmmtools_amd_seed', np.array([int(data['seed'])], dtype=np.int64))
            flat = dict(number_of_iterations=data.get('number_of_iterations'))
            flat.update(data['kwargs'])
            data = {k: (v.tolist() if isinstance(v, np.ndarray) else v) for k, v in flat.items()}
        if name == 'metadata':
            # :1127-1139: the title is a global attribute, the rest is stored nested (a group per dictionary, a fixed-length
            # character variable per value) so that one entry can be read without the rest
            if data.get('title') is not None:
                self._a.set_attr('/', 'title', str(data.pop('title')))
            else:
                data.pop('title', None)
            self._write_nested('/metadata', data)
            return
        self._write_text(self._a, '/' + name.strip('/'), _yaml_dump(data), fixed=bool(fixed_dimension))

    def _write_nested(self, path, value, fixed=True):
        if isinstance(value, dict) and len(value) > 0:
            for k, v in value.items():
                if not isinstance(k, str):
                    raise ValueError('Cannot store dict in nested form with non-string keys.')          # :1854-1856
                self._write_nested(path + '/' + k, v, fixed)
            return
        self._write_text(self._a, path, _yaml_dump(value), fixed=fixed)

    # ---- states (:612-668) -------------------------------------------------------------------------------------
    def write_thermodynamic_states(self, thermodynamic_states, unsampled_states):
        from .. import system_xml
        import zlib
        first_of = []                                   # (state, 'kind/index') of the first state of every compatible group
        for kind, states in (('thermodynamic_states', thermodynamic_states), ('unsampled_states', unsampled_states)):
            for k, s in enumerate(states):
                name = type(s).__name__
                if name not in ('ThermodynamicState', 'CompoundThermodynamicState'):
                    raise NotImplementedError('%s in the reference\'s store layout' % name)
                d = {'_serialized__class_name': 'ThermodynamicState', '_serialized__module_name': 'openmmtools.states',
                     'temperature': _Quantity(float(s.temperature), 'kelvin'), 'surface_tension': None,
                     'pressure': None if s.pressure is None else _Quantity(float(s.pressure) / _UNIT_TO_MD['bar'], 'bar')}
                ref = next((name for other, name in first_of if s.is_state_compatible(other)), None)
                if ref is None:
                    xml = system_xml.to_xml(s.system, pressure=None if s.pressure is None else _STANDARD_PRESSURE_BAR * _UNIT_TO_MD['bar'],
                                            temperature=_STANDARD_TEMPERATURE)
                    d['standard_system'] = zlib.compress(xml.encode())
                    first_of.append((s, '%s/%d' % (kind, k)))
                else:
                    d['_Reporter__compatible_state'] = ref
                if name == 'CompoundThermodynamicState':
                    # states.py:2956-2971 around the plain state; the AlchemicalState as GlobalParameterState.__getstate__
                    # writes it (:3879-3898): the lambdas the System defines, None for the ones it does not (alchemy.py:94-99)
                    # (one entry per composable state: the region's name as parameters_name_suffix, the parameters under their plain names)
                    alch = [{'_serialized__class_name': 'AlchemicalState', '_serialized__module_name': 'openmmtools.alchemy.alchemy',
                             'parameters': {'lambda_sterics': float(c.lambda_sterics), 'lambda_electrostatics': float(c.lambda_electrostatics),
                                            **{k: (float(getattr(c, k)) if k in c._defined else None) for k in ('lambda_bonds', 'lambda_angles', 'lambda_torsions')}},
                             'function_variables': {}, 'parameters_name_suffix': c.parameters_name_suffix} for c in s._alchs]
                    d = {'_serialized__class_name': 'CompoundThermodynamicState', '_serialized__module_name': 'openmmtools.states',
                         'thermodynamic_state': d, 'composable_states': alch}
                self._write_text(self._a, '/%s/state%d' % (kind, k), _yaml_dump(d), fixed=True)

    def write_mcmc_moves(self, mcmc_moves):
        """:813-815 (one variable per state's move, rewritten every iteration by the reference; the recipes here are fixed)."""
        for k, m in enumerate(mcmc_moves):
            name = type(m).__name__
            if name not in ('LangevinSplittingDynamicsMove', 'LangevinDynamicsMove'):
                raise NotImplementedError('MCMC move %s in the reference\'s store layout' % name)
            d = {'_serialized__class_name': name, '_serialized__module_name': 'openmmtools.mcmc',
                 'timestep': _Quantity(float(m.timestep) * 1e3, 'femtosecond'),
                 'collision_rate': _Quantity(float(m.collision_rate), '/picosecond'),
                 'n_steps': int(m.n_steps), 'reassign_velocities': bool(m.reassign_velocities),
                 'constraint_tolerance': float(getattr(m, 'constraint_tolerance', 1e-8)),
                 'n_restart_attempts': int(getattr(m, 'n_restart_attempts', 4))}
            if name == 'LangevinSplittingDynamicsMove':
                d.update(splitting=str(m.splitting), measure_heat=bool(getattr(m, 'measure_heat', False)),
                         measure_shadow_work=bool(getattr(m, 'measure_shadow_work', False)))
            path = '/mcmc_moves/move%d' % k
            text = _yaml_dump(d)
            if path in self._a and str(np.asarray(self._a.read(path)).reshape(-1)[0]) == text:
                continue
            self._write_text(self._a, path, text, fixed=False)

    # ---- per-iteration variables (:775-1018) -------------------------------------------------------------------
    def _record_variable(self, f, path, kind, dims, sizes, attrs=()):
        if path in f:
            return
        group = path.rsplit('/', 1)[0]
        if group:
            f.create_group(group)
        for d, n in zip(dims, sizes):
            where = (group + '/' + d) if (group and d.startswith('dim_size')) else '/' + d
            if n is not None and not f.has_dimension(where):
                f.create_dimension(where, n)
        f.create_variable(path, kind, dims)
        for k, v in attrs:
            f.set_attr(path, k, v)

    def write_replica_thermodynamic_states(self, state_indices, iteration):
        s = np.asarray(state_indices)
        self._record_variable(self._a, '/states', 'i4', ('iteration', 'replica'), (None, len(s)),
                              (('units', 'none'), ('long_name', "states[iteration][replica] is the thermodynamic state index (0..n_states-1) "
                                                                "of replica 'replica' of iteration 'iteration'.")))
        self._a.write('/states', s, record=int(iteration))

    def write_energies(self, energy_thermodynamic_states, energy_neighborhoods, energy_unsampled_states, iteration):
        e = np.asarray(energy_thermodynamic_states)
        R, K = e.shape
        self._record_variable(self._a, '/energies', 'f8', ('iteration', 'replica', 'state'), (None, R, K),
                              (('units', 'kT'), ('long_name', "energies[iteration][replica][state] is the reduced (unitless) energy of "
                                                             "replica 'replica' from iteration 'iteration' evaluated at the thermodynamic state 'state'.")))
        self._record_variable(self._a, '/neighborhoods', 'i1', ('iteration', 'replica', 'state'), (None, R, K),
                              (('_FillValue', np.array([1], dtype=np.int8)),            # :901: old files read as "all states"
                               ('long_name', "neighborhoods[iteration][replica][state] is 1 if this energy was computed during this iteration.")))
        self._a.write('/energies', e, record=int(iteration))
        self._a.write('/neighborhoods', np.asarray(energy_neighborhoods), record=int(iteration))
        eu = np.asarray(energy_unsampled_states)
        if eu.size:
            self._record_variable(self._a, '/unsampled_energies', 'f8', ('iteration', 'replica', 'unsampled'), (None, R, eu.shape[1]),
                                  (('units', 'kT'), ('long_name', "unsampled_energies[iteration][replica][state] is the reduced (unitless) energy of replica "
                                                                 "'replica' from iteration 'iteration' evaluated at unsampled thermodynamic state 'state'.")))
            self._a.write('/unsampled_energies', eu, record=int(iteration))

    def write_mixing_statistics(self, n_accepted_matrix, n_proposed_matrix, iteration):
        a = np.asarray(n_accepted_matrix)
        for name, m, what in (('accepted', a, 'accepted'), ('proposed', np.asarray(n_proposed_matrix), 'proposed')):
            self._record_variable(self._a, '/' + name, 'i4', ('iteration', 'state', 'state'), (None, a.shape[0], a.shape[0]),
                                  (('units', 'none'), ('long_name', "%s[iteration][i][j] is the number of %s transitions between states i and j "
                                                                    "from iteration 'iteration-1'." % (name, what))))
            self._a.write('/' + name, m, record=int(iteration))

    def write_timestamp(self, iteration):
        import time
        self._record_variable(self._a, '/timestamp', 'str', ('iteration',), (None,))
        self._a.write('/timestamp', time.ctime(), record=int(iteration))

    def write_last_iteration(self, iteration):
        self._a.write('/last_iteration', [int(iteration)])
        self._c.write('/last_iteration', [int(iteration)])
        self.sync()

    def write_online_analysis(self, iteration, **kwargs):
        """:1167-1252: numeric arrays under online_analysis/: '<name>' (latest) and '<name>_history' (per iteration)."""
        kwargs = dict(kwargs)
        sams = kwargs.pop('sams_state', None)
        if isinstance(sams, dict):
            # sams.py:615-620: stage and t0 are stored beside logZ; the state histogram (recomputed from the stored states by the
            # reference, :437-451) is kept as well so that a resume here does not have to re-read every iteration
            kwargs.update(stage=sams['stage'], t0=sams['t0'], state_histogram=sams['histogram'])
        for name, v in kwargs.items():
            if v is None:
                continue
            try:
                arr = np.atleast_1d(np.asarray(v, dtype=np.float64))
            except (TypeError, ValueError):
                continue
            if arr.ndim != 1:
                continue
            dim = 'dim_size%d' % arr.size
            self._record_variable(self._a, '/online_analysis/' + name, 'f8', (dim,), (arr.size,))
            self._record_variable(self._a, '/online_analysis/%s_history' % name, 'f8', ('iteration', dim), (None, arr.size))
            self._a.write('/online_analysis/' + name, arr)
            self._a.write('/online_analysis/%s_history' % name, arr, record=int(iteration))

    # ---- checkpoints (:1597-1737) ------------------------------------------------------------------------------
    def write_sampler_states(self, sampler_states, iteration):
        if self._analysis_particles:
            # :722-741: the chosen particles go to the ANALYSIS file every iteration (one frame per iteration there)
            sel = list(self._analysis_particles)
            xs = np.stack([np.asarray(s.positions, dtype=np.float64)[sel] for s in sampler_states])
            vs = np.stack([np.zeros((len(sel), 3)) if s.velocities is None else np.asarray(s.velocities, dtype=np.float64)[sel] for s in sampler_states])
            R = xs.shape[0]
            a = self._a
            self._record_variable(a, '/positions', 'f4', ('iteration', 'replica', 'analysis_particles', 'spatial'), (None, R, None, 3), (('units', 'nm'),))
            self._record_variable(a, '/velocities', 'f4', ('iteration', 'replica', 'analysis_particles', 'spatial'), (None, R, None, 3), (('units', 'nm / ps'),))
            # :1686-1692: frames of the analysis trajectory only every position_interval / velocity_interval iterations (0: never);
            # the iterations in between stay at the fill value, as records netCDF never wrote do
            if self._position_interval != 0 and int(iteration) % self._position_interval == 0:
                a.write('/positions', xs, record=int(iteration))
                if all(s.box_vectors is not None for s in sampler_states):                      # :1720-1731: boxes go with positions
                    box = np.stack([np.asarray(s.box_vectors, dtype=np.float64).reshape(3, 3) for s in sampler_states])
                    self._record_variable(a, '/box_vectors', 'f4', ('iteration', 'replica', 'spatial', 'spatial'), (None, R, 3, 3), (('units', 'nm'),))
                    self._record_variable(a, '/volumes', 'f8', ('iteration', 'replica'), (None, R), (('units', 'nm**3'),))
                    a.write('/box_vectors', box, record=int(iteration))
                    a.write('/volumes', np.abs(np.linalg.det(box)), record=int(iteration))
            if self._velocity_interval != 0 and int(iteration) % self._velocity_interval == 0:
                a.write('/velocities', vs, record=int(iteration))
        if iteration % self._interval != 0:
            return False
        frame = int(iteration) // self._interval
        x = np.stack([np.asarray(s.positions, dtype=np.float64) for s in sampler_states])
        R, N = x.shape[0], x.shape[1]
        v = np.stack([np.zeros((N, 3)) if s.velocities is None else np.asarray(s.velocities, dtype=np.float64) for s in sampler_states])
        c = self._c
        self._record_variable(c, '/positions', 'f4', ('iteration', 'replica', 'atom', 'spatial'), (None, R, N, 3),
                              (('units', 'nm'), ('long_name', "positions[iteration][replica][atom][spatial] is position of coordinate 'spatial' "
                                                             "of atom 'atom' from replica 'replica' for iteration 'iteration'.")))
        self._record_variable(c, '/velocities', 'f4', ('iteration', 'replica', 'atom', 'spatial'), (None, R, N, 3),
                              (('units', 'nm / ps'), ('long_name', "velocities[iteration][replica][atom][spatial] is velocity of coordinate 'spatial' "
                                                                  "of atom 'atom' from replica 'replica' for iteration 'iteration'.")))
        c.write('/positions', x, record=frame)
        c.write('/velocities', v, record=frame)
        if all(s.box_vectors is not None for s in sampler_states):
            box = np.stack([np.asarray(s.box_vectors, dtype=np.float64).reshape(3, 3) for s in sampler_states])
            self._record_variable(c, '/box_vectors', 'f4', ('iteration', 'replica', 'spatial', 'spatial'), (None, R, 3, 3),
                                  (('units', 'nm'), ('long_name', "box_vectors[iteration][replica][i][j] is dimension j of box vector i for replica "
                                                                 "'replica' from iteration 'iteration-1'.")))
            self._record_variable(c, '/volumes', 'f8', ('iteration', 'replica'), (None, R),
                                  (('units', 'nm**3'), ('long_name', "volume[iteration][replica] is the box volume for replica 'replica' "
                                                                    "from iteration 'iteration-1'.")))
            c.write('/box_vectors', box, record=frame)
            c.write('/volumes', np.abs(np.linalg.det(box)), record=frame)
        return True

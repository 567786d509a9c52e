"""One engine handle per group of compatible thermodynamic states, behind the interface of a single engine.

The reference keeps one OpenMM Context per *compatible* group of states (same standard System: ``states.group_by_compatibility``
states.py:186-217, ``is_state_compatible`` :994-1050) in ``cache.ContextCache``; a replica is propagated in the Context of its own
state's System (multistatesampler.py:1296-1320) and its configuration is evaluated in one Context per group for the energy
matrix (:1470-1490).  Here a group is an engine handle with its own System tables on the device:

    propagation   the local replicas are partitioned by the group of their current state; each group's handle receives its
                  share as one batch, propagates it and hands it back -- batches, not one launch chain per replica; a group
                  without replicas in an iteration is skipped
    u_kl          a second handle per group holds ALL local replicas and fills the columns of the group's states
    mixing        any handle: the swap kernels only see the energy matrix

Round 4: the master copy of positions / velocities / boxes lives ON THE DEVICE, in the first group's all-replica handle; shares
go to the propagation handles and back, and the full set to the other groups' energy handles, through ``copy_replicas``
(remd_copy_replicas: device-to-device rows, no host staging; the reference moves them between Contexts through
SamplerState.apply_to_context).  The host sees coordinates only in ``set_replicas`` / ``get_replicas``.  Random streams: every handle carries the ensemble's seed and a group's share of
the local replicas is keyed by the replicas' GLOBAL indices (``set_replica_ids`` -> remd_set_replica_ids), exactly as a
single-group run keys its block by r_begin + r: velocities, Langevin noise and Metropolis draws of a replica do not depend on
which handle runs it, so a multi-group run is independent of the rank count too (tests/test_compat_groups.py; round 4 -- before,
handles were seeded per (seed, group, rank offset)).  The Monte Carlo barostat's attempt counter is per handle and still is not.
"""
import numpy as np


class EnginePool:
    is_device = False          # rows and labels pass through the host (the sampler takes its host all-gather path); coordinates do not

    def __init__(self, first_engine, state_groups):
        """first_engine: an engine of the kind to pool (its ``spawn()`` or its class makes the others).
        state_groups: list over groups of the global state indices (sampled + unsampled) in that group."""
        self._first = first_engine
        self._groups = [list(map(int, g)) for g in state_groups]
        self.G = len(self._groups)
        self.K = sum(len(g) for g in self._groups)
        self._group_of_state = np.zeros(self.K, dtype=np.int64)
        self._local_index = np.zeros(self.K, dtype=np.int64)
        for g, idx in enumerate(self._groups):
            for l, k in enumerate(idx):
                self._group_of_state[k], self._local_index[k] = g, l
        self.ewald_split = getattr(first_engine, 'ewald_split', None)
        self._energy = [first_engine] + [self._spawn() for _ in range(self.G - 1)]     # all local replicas, u_kl columns
        self._prop = [None] * self.G                                                   # made when a group first holds a replica
        self._setup = []                                                               # calls replayed on a late propagation handle
        self._descs = None
        self._seed = 0
        self.device = getattr(first_engine, 'device', 0)

    def _spawn(self):
        return self._first.spawn() if hasattr(self._first, 'spawn') else type(self._first)()

    # ---- configuration: forwarded to every handle, recorded for the ones made later ---------------------------------
    def _each(self, name, per_group_args):
        self._setup = [c for c in self._setup if c[0] != name] + [(name, per_group_args)]
        for g in range(self.G):
            for eng in (self._energy[g], self._prop[g]):
                if eng is not None and hasattr(eng, name):
                    getattr(eng, name)(*per_group_args(g))

    def set_system(self, descs):
        if len(descs) != self.G:
            raise ValueError('one system description per compatibility group')
        n = {int(d['n_atoms']) for d in descs}
        if len(n) != 1:
            raise ValueError('the Systems of all thermodynamic states must hold the same particles (%s)' % sorted(n))
        self._descs = list(descs)
        self.N = n.pop()
        self._each('set_system', lambda g: (self._descs[g],))
        for eng in self._prop:                     # (a handle that got a new System is sized again before its labels-only fast path)
            if eng is not None:
                eng._pool_size = None

    def set_states(self, beta, lambda_sterics=None, lambda_electrostatics=None, energy_const=None):
        def pick(a, g):
            return None if a is None else np.asarray(a, dtype=np.float64)[self._groups[g]]
        self._each('set_states', lambda g: (pick(beta, g), pick(lambda_sterics, g), pick(lambda_electrostatics, g),
                                            pick(energy_const, g)))

    def set_integrator(self, *args):
        self._each('set_integrator', lambda g: args)

    def set_restart_attempts(self, n):
        self._each('set_restart_attempts', lambda g: (n,))

    def set_work_measurement(self, measure_heat=False, measure_shadow_work=False):
        self._each('set_work_measurement', lambda g: (measure_heat, measure_shadow_work))

    def set_barostat(self, pressure, frequency=25):
        self._each('set_barostat', lambda g: (None if pressure is None else np.asarray(pressure, dtype=np.float64)[self._groups[g]],
                                              frequency))

    def set_energy_const_volume(self, volume):
        self._each('set_energy_const_volume', lambda g: (volume,))

    def seed(self, seed):
        self._seed = int(seed)
        for g in range(self.G):
            self._energy[g].seed(seed)                     # (mixing draws from the first handle: the sampler's own stream)
            if self._prop[g] is not None:
                self._prop[g].seed(self._seed)

    def _propagator(self, g):
        if self._prop[g] is None:
            eng = self._spawn()
            for name, args in self._setup:
                if hasattr(eng, name):
                    getattr(eng, name)(*args(g))
            eng.seed(self._seed)
            self._prop[g] = eng
        return self._prop[g]

    # ---- replicas: master copy on the host ---------------------------------------------------------------------------
    def set_replicas(self, R_global, r_begin, x, v, box, labels):
        self.R_global, self.r_begin = int(R_global), int(r_begin)
        x = np.asarray(x, dtype=np.float64)
        self.R = x.shape[0]
        box = np.array(box, dtype=np.float64).reshape(self.R, 3)
        zero = np.zeros(self.R, dtype=np.int64)
        self._master.set_replicas(self.R, 0, x, v, box, zero)           # the master copy (labels of a handle are group-local)
        for eng in self._energy[1:]:
            eng.set_replicas(self.R, 0, None, None, box, zero)          # sized here, filled by copy_replicas before each use
        self.reset_work()
        self.set_labels(labels)

    @property
    def _master(self):
        return self._energy[0]

    def set_labels(self, labels):
        self._labels = np.array(labels, dtype=np.int64)

    def get_replicas(self, positions=True, velocities=True, potential=False, kinetic=False):
        if potential or kinetic:
            raise NotImplementedError('per-replica energies are not kept across compatibility groups')
        x, v, _, _ = self._master.get_replicas(positions, velocities)
        return x, v, None, None

    def get_boxes(self):
        return self._master.get_boxes()

    def _local_groups(self):
        mine = self._labels[self.r_begin:self.r_begin + self.R]
        return self._group_of_state[mine], self._local_index[mine]

    def _on_groups(self, call, periodic_boxes):
        """Run ``call(engine) -> per-replica result or None`` on each group's share of the local replicas."""
        grp, loc = self._local_groups()
        out = {}
        boxes = None
        for g in range(self.G):
            idx = np.nonzero(grp == g)[0]
            if len(idx) == 0:
                continue
            eng = self._propagator(g)
            slots = np.arange(len(idx))
            if getattr(eng, '_pool_size', None) == len(idx) and hasattr(eng, 'set_labels'):
                # the handle already holds this many replicas: labels only (no host-built lattice, no upload, no re-allocation of
                # the mesh buffers) -- the coordinates and boxes follow device to device below
                eng.set_labels(loc[idx])
            else:
                if boxes is None:
                    boxes = self._master.get_boxes()                 # (3 numbers per replica: only to size the handle)
                eng.set_replicas(len(idx), 0, None, None, boxes[idx], loc[idx])
                eng._pool_size = len(idx)
            eng.copy_replicas(slots, self._master, idx, 7)           # positions, velocities, boxes: device to device
            eng.set_replica_ids(self.r_begin + idx)                  # noise keyed by the global replica index, whatever the grouping
            if hasattr(eng, 'reset_work'):
                eng.reset_work()                                     # (a handle's accumulators are per slot; the pool's per replica)
            out[g] = (idx, call(eng))
            if hasattr(eng, 'get_work'):
                w = eng.get_work()
                for key in self._work:
                    self._work[key][idx] += w[key]
            self._master.copy_replicas(idx, eng, slots, 7 if periodic_boxes else 3)
        return out

    def propagate(self, iteration):
        flags = np.zeros(self.R, dtype=np.int32)
        for idx, f in self._on_groups(lambda eng: eng.propagate(iteration), self._has_barostat()).values():
            flags[idx] = f
        return flags

    def barostat_attempts(self, n_attempts):
        self._on_groups(lambda eng: eng.barostat_attempts(n_attempts), True)

    def minimize(self, tolerance=1.0, max_iterations=0):
        conv = np.zeros(self.R, dtype=np.int32)
        steps = 0
        for idx, (c, n) in self._on_groups(lambda eng: eng.minimize(tolerance, max_iterations), False).values():
            conv[idx] = c
            steps = max(steps, int(n))
        return conv, steps

    def _has_barostat(self):
        for name, args in self._setup:
            if name == 'set_barostat':
                return args(0)[0] is not None
        return False

    # ---- energies and mixing -----------------------------------------------------------------------------------------
    def compute_energies(self, d_rows=None, want_host=True, want_potential=False):
        if d_rows is not None or want_potential:
            raise NotImplementedError('the pooled engines return host rows only')
        rows = np.empty((self.R, self.K), dtype=np.float64)
        every = np.arange(self.R)
        for g in range(self.G):
            eng = self._energy[g]
            if g > 0:
                eng.copy_replicas(every, self._master, every, 5)     # positions and boxes of all local replicas
            rows[:, self._groups[g]] = eng.compute_energies()
        self._rows = rows
        return rows

    def mix_host(self, *args, **kwargs):
        return self._first.mix_host(*args, **kwargs)

    def mix(self, scheme, iteration, labels, d_ukl=None, R=None, K=None, ld=0, log_weights=None):
        """Single-process mixing on the rows of the last ``compute_energies`` (the first K columns: the sampled states)."""
        if d_ukl is not None:
            raise NotImplementedError('the pooled engines mix from host rows')
        K = self.K if K is None else K
        return self._first.mix_host(scheme, iteration, np.ascontiguousarray(self._rows[:, :K]), labels, log_weights=log_weights)

    def get_work(self):
        """Heat, shadow work and Metropolis counters per local replica, summed over the handles that propagated it."""
        return {k: v.copy() for k, v in self._work.items()}

    def reset_work(self):
        self._work = dict(heat=np.zeros(self.R), shadow_work=np.zeros(self.R), n_accepted=np.zeros(self.R, np.int64),
                          n_trials=np.zeros(self.R, np.int64))

    def close(self):
        for eng in self._energy + self._prop:
            if eng is not None and hasattr(eng, 'close'):
                eng.close()

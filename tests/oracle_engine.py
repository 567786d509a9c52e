"""OracleEngine: the Python engine interface (openmmtools_amd._engine.HipEngine) backed by the CPU
oracle.  TEST INFRASTRUCTURE ONLY — lets the host-side sampler logic, the sharding and the
collectives be exercised on a GPU-less box (gloo, world_size 2) and serves as the checker in the
GPU parity tests.  Never imported by the product package."""
import numpy as np
import oracle
from oracle import md_oracle as mo


class OracleEngine:
    is_device = False

    def __init__(self, system_factory=None):
        self.system_factory = system_factory or mo.OracleSystem
        self.seed_value = 0

    def spawn(self):
        return type(self)(self.system_factory)

    def set_system(self, desc):
        self.desc = desc
        self.sys = self.system_factory(desc)
        self.N = self.sys.N

    def set_states(self, beta, lambda_sterics=None, lambda_electrostatics=None, energy_const=None):
        self.beta = np.array(beta, dtype=np.float64)
        self.K = len(self.beta)
        self.lam_s = np.ones(self.K) if lambda_sterics is None else np.array(lambda_sterics, dtype=np.float64)
        self.lam_e = np.ones(self.K) if lambda_electrostatics is None else np.array(lambda_electrostatics, dtype=np.float64)
        self.econst = np.zeros(self.K) if energy_const is None else np.array(energy_const, dtype=np.float64)

    def set_integrator(self, splitting, timestep, collision_rate, n_steps, reassign_velocities=True,
                       constraint_tolerance=1e-8):
        self.integ_args = (splitting, timestep, collision_rate, n_steps)
        self.reassign = reassign_velocities

    def set_barostat(self, pressure, frequency=25):
        self.pressure = None if pressure is None else np.array(pressure, dtype=np.float64)
        self.baro_frequency = int(frequency)
        self._baro = None
        self._baro_steps = 0
        self._baro_attempts = 0

    def set_energy_const_volume(self, volume):
        self.econst_vref = float(volume)

    def get_boxes(self):
        return self.box.copy()

    def set_restart_attempts(self, n):
        self.n_restart_attempts = int(n)

    # heat / shadow work / Metropolization (remd_set_work_measurement, remd_get_work, remd_reset_work)
    def set_work_measurement(self, measure_heat=False, measure_shadow_work=False):
        self._measure = (bool(measure_heat), bool(measure_shadow_work))

    def _work_of(self, integ, r, tokens=None):
        toks = integ.tokens if tokens is None else tokens
        on = any(getattr(self, '_measure', (False, False))) or ('}' in toks)
        if not on:
            integ.work = None
            return
        if getattr(self, '_work', None) is None or len(self._work) != self.R:
            self._work = [dict(heat=0.0, shadow_work=0.0, n_accepted=0, n_trials=0) for _ in range(self.R)]
        integ.work = self._work[r]

    def get_work(self):
        w = getattr(self, '_work', None) or [dict(heat=0.0, shadow_work=0.0, n_accepted=0, n_trials=0) for _ in range(self.R)]
        m = getattr(self, '_measure', (False, False))
        return dict(heat=np.array([x['heat'] if m[0] else 0.0 for x in w]), shadow_work=np.array([x['shadow_work'] for x in w]),
                    n_accepted=np.array([x['n_accepted'] for x in w], np.int64), n_trials=np.array([x['n_trials'] for x in w], np.int64))

    def reset_work(self):
        self._work = None

    def set_replicas(self, R_global, r_begin, x, v, box, labels):
        self.R_global, self.r_begin = R_global, r_begin
        self.box = np.array(box, dtype=np.float64).reshape(-1, 3)
        # (x = None: the handle is only sized, coordinates follow through copy_replicas -- include/remd_hip.h)
        self.x = np.zeros((len(self.box), self.N, 3)) if x is None else np.array(x, dtype=np.float64)
        self.R = self.x.shape[0]
        self.v = np.zeros_like(self.x) if v is None else np.array(v, dtype=np.float64)
        self.labels = np.array(labels, dtype=np.int64)
        self.noise_ids = None                     # (ids belong to one set of replicas: include/remd_hip.h remd_set_replica_ids)

    def set_replica_ids(self, ids):
        self.noise_ids = None if ids is None else np.array(ids, dtype=np.int64)

    def copy_replicas(self, slots, source, source_slots, what=7):
        """remd_copy_replicas: positions (1), velocities (2), boxes (4) of ``source``'s slots into this engine's slots."""
        d, s = np.asarray(slots, dtype=np.int64), np.asarray(source_slots, dtype=np.int64)
        if what & 1:
            self.x[d] = source.x[s]
        if what & 2:
            self.v[d] = source.v[s]
        if what & 4:
            self.box[d] = source.box[s]

    def _nk(self, r):
        ids = getattr(self, 'noise_ids', None)
        return int(self.r_begin + r) if ids is None else int(ids[r])

    def set_labels(self, labels):
        self.labels = np.array(labels, dtype=np.int64)

    def seed(self, seed):
        self.seed_value = int(seed)

    def _integrator(self):
        s, dt, g, n = self.integ_args
        return mo.OracleLangevin(self.sys, s, dt, g, n, self.seed_value)

    def _box(self, r):
        return self.box[r] if self.box[r].any() else None

    def propagate(self, iteration):
        integ = self._integrator()
        flags = np.zeros(self.R, dtype=np.int32)
        for r in range(self.R):
            rg = self.r_begin + r
            k = self.labels[rg]
            kT = 1.0 / self.beta[k]
            x0, v0 = self.x[r].copy(), self.v[r].copy()
            # mcmc.py:706-759: a NaN result restores the start state and repeats the move with fresh noise (the attempt
            # number rides in the high bits of the iteration counter, as in remd_propagate)
            for attempt in range(getattr(self, 'n_restart_attempts', 0) + 1):
                it = iteration + (attempt << 40)
                v = integ.assign_velocities(x0, kT, self._nk(r), it) if self.reassign else v0
                self._work_of(integ, r)
                if integ.work is not None:                # (what a discarded attempt accumulated for get_work goes with it)
                    if attempt == 0:
                        work0 = dict(integ.work)
                    else:
                        integ.work.update(work0)
                if getattr(self, 'pressure', None) is not None:
                    if self._baro is None:
                        self._baro = mo.OracleBarostat(self.sys, self.seed_value, mo.molecules_from_desc(self.sys.d))
                    vref = getattr(self, 'econst_vref', 0.0)
                    baro = dict(obj=self._baro, pressure=self.pressure[k], frequency=self.baro_frequency,
                                steps_done=self._baro_steps, attempts_done=self._baro_attempts,
                                long_range=(self.econst[k] * vref) if vref > 0 else 0.0)
                    self.x[r], self.v[r], self.box[r] = integ.run(x0, v, self._box(r), kT, self._nk(r), it, lambda_sterics=self.lam_s[k],
                                                                  lambda_electrostatics=self.lam_e[k], barostat=baro)
                else:
                    self.x[r], self.v[r] = integ.run(x0, v, self._box(r), kT, self._nk(r), it,
                                                     lambda_sterics=self.lam_s[k], lambda_electrostatics=self.lam_e[k])
                flags[r] = 0 if (np.isfinite(self.x[r]).all() and np.isfinite(self.v[r]).all()) else 1
                if not flags[r]:
                    break
        if getattr(self, 'pressure', None) is not None:
            n = integ.n_steps
            self._baro_attempts += (self._baro_steps + n) // self.baro_frequency - self._baro_steps // self.baro_frequency
            self._baro_steps += n
        return flags

    def barostat_attempts(self, n_attempts):
        """remd_barostat_attempts: n volume moves per replica outside the integrator, same attempt counter."""
        if getattr(self, 'pressure', None) is None:
            raise RuntimeError('remd_barostat_attempts: no barostat (remd_set_barostat)')
        if self._baro is None:
            self._baro = mo.OracleBarostat(self.sys, self.seed_value, mo.molecules_from_desc(self.sys.d))
        vref = getattr(self, 'econst_vref', 0.0)
        for a in range(int(n_attempts)):
            for r in range(self.R):
                rg = self.r_begin + r
                k = self.labels[rg]
                lr = (self.econst[k] * vref) if vref > 0 else 0.0
                self.x[r], self.box[r], _ = self._baro.attempt(self.x[r], self._box(r), 1.0 / self.beta[k], self.pressure[k], self._nk(r),
                                                               self._baro_attempts, long_range=lr,
                                                               lambda_sterics=self.lam_s[k], lambda_electrostatics=self.lam_e[k])
            self._baro_attempts += 1

    def minimize(self, tolerance=1.0, max_iterations=0):
        fire = mo.OracleFIRE(self.sys, tolerance=tolerance)
        conv = np.zeros(self.R, dtype=np.int32)
        n_it = 0
        for r in range(self.R):
            k = self.labels[self.r_begin + r]
            self.x[r], self.v[r], _, c, it = fire.minimize(self.x[r], self._box(r), max_iterations,
                                                           lambda_sterics=self.lam_s[k], lambda_electrostatics=self.lam_e[k])
            conv[r] = int(c)
            n_it = max(n_it, it)
        return conv, n_it

    def step(self, splitting, iteration=0, first_step=0, n_steps=1):
        integ = self._integrator()
        for r in range(self.R):
            rg = self.r_begin + r
            k = self.labels[rg]
            self._work_of(integ, r, [c for c in splitting.upper() if c != ' '])
            self.x[r], self.v[r] = integ.run(self.x[r], self.v[r], self._box(r), 1.0 / self.beta[k], self._nk(r), iteration,
                                             first_step=first_step, n_steps=n_steps,
                                             tokens=[c for c in splitting.upper() if c != ' '],
                                             lambda_sterics=self.lam_s[k], lambda_electrostatics=self.lam_e[k])

    def potentials(self):
        """The potential energy of each replica in its OWN thermodynamic state (what the reference's SamplerState carries)."""
        out = []
        for r in range(self.R):
            k = self.labels[self.r_begin + r]
            lam = {}
            if hasattr(self.sys, 'state_energies'):
                lam = dict(lambda_sterics=self.lam_s[k], lambda_electrostatics=self.lam_e[k])
            out.append(self.sys.potential(self.x[r], self._box(r), **lam))
        return np.array(out)

    def compute_energies(self, d_rows=None, want_host=True, want_potential=False):
        U = self.potentials()
        vref = getattr(self, 'econst_vref', 0.0)
        scale = [(vref / np.prod(self.box[r])) if vref > 0 else 1.0 for r in range(self.R)]      # constants ~ 1/V
        if hasattr(self.sys, 'state_energies'):
            rows = np.stack([self.beta * (self.sys.state_energies(self.x[r], self._box(r), self.lam_s, self.lam_e)
                                          + self.econst * scale[r]) for r in range(self.R)])
        else:
            rows = np.stack([self.beta * (U[r] + self.econst * scale[r]) for r in range(self.R)])
        if getattr(self, 'pressure', None) is not None:                        # states.py:1913-1914: + beta_l p_l V_r
            rows = rows + self.beta[None, :] * self.pressure[None, :] * np.prod(self.box, axis=1)[:, None]
        self._rows = rows
        return (rows, U) if want_potential else rows

    def mix(self, scheme, iteration, labels, d_ukl=None, R=None, K=None, ld=0, log_weights=None):
        # single-rank path: the engine's own rows are the full matrix
        return self.mix_host(scheme, iteration, self._rows[:, :(K or self.K)], labels, log_weights=log_weights)

    def mix_host(self, scheme, iteration, ukl, labels, log_weights=None, n_attempts=-1):
        return oracle.mix(scheme, self.seed_value, iteration, ukl, labels, log_weights=log_weights, n_attempts=n_attempts)

    def get_replicas(self, positions=True, velocities=True, potential=False, kinetic=False):
        u = self.potentials() if potential else None
        k = np.array([mo.kinetic_energy(self.sys.mass, self.v[r]) for r in range(self.R)]) if kinetic else None
        return self.x.copy(), self.v.copy(), u, k

    def get_forces(self):
        return np.stack([self.sys.energy_forces(self.x[r], self._box(r))[1] for r in range(self.R)])

    def sync(self):
        pass

    def close(self):
        pass

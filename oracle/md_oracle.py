"""md_oracle.py — TEST INFRASTRUCTURE ONLY (CPU oracle, f64 numpy).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module;
the product path (openmmtools_amd/_engine.py -> libremd_hip.so) never does.

Restates, in double precision and the simplest possible form, the arithmetic the hot path
performs per replica:

  * Langevin splitting substeps   openmmtools/integrators.py:1404-1460 (R, V, O), constants
                                  :1139-1149, sigma = sqrt(kT/m) :1314, step program :1309-1317
  * velocity reassignment         openmmtools/mcmc.py:710-711
  * reduced potential             openmmtools/states.py:1908-1917; PT shortcut
                                  openmmtools/multistate/paralleltempering.py:206-215
  * potential energy / forces of the benchmark systems as OpenMM defines them for the Systems
    built in openmmtools/testsystems.py:779-786 (harmonic well), :1957-2017 (LJ fluid with
    switching function and dispersion correction), :3496-3527 (Amber explicit solvent: bonds,
    angles, torsions, LJ + PME Coulomb with exclusions/exceptions), and the alchemical soft-core
    forms of openmmtools/alchemy/alchemy.py:1356-1390.
  * constraints by plain iterative SHAKE / RATTLE to 1e-12 (the device uses analytic SETTLE for
    waters, so agreement is a real cross-check, not the same code twice).

PARITY PINNING: the force/energy arithmetic lives in OpenMM (not vendored in /root/reference,
CI pins 8.2.0 / 8.3.1, .github/workflows/CI.yml:31-43) and no reference test pins an absolute
energy or trajectory, so absolute-energy parity is UNPINNED against the reference.
What pins this oracle: closed forms (harmonic well, two-particle LJ, analytic long-range
correction integral), direct Ewald summation for the PME path, finite-difference force checks,
and the statistical known answers the reference tests use (tests/test_oracle_md.py).

Random numbers: Philox4x32-10 with the stream layout of openmmtools_amd/csrc/rng.h, restated
here with numpy uint64 arithmetic and checked against the Random123 known-answer vectors.
"""
import math
import numpy as np

KB = 0.008314462618153242          # kJ/mol/K  (openmmtools/constants.py:7)
ONE_4PI_EPS0 = 138.93545764438198  # openmmtools/constants.py:12-14

STREAM_SWAP_ALL, STREAM_NEIGHBOR, STREAM_SAMS, STREAM_VELOCITY, STREAM_OU = 1, 2, 3, 4, 5
STREAM_METROPOLIS = 7          # a = index of the '}' in the step program, b = global replica, t = global step; uniform from words 2, 3

_M0, _M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
_W0, _W1 = 0x9E3779B9, 0xBB67AE85
_MASK = np.uint64(0xFFFFFFFF)


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Vectorised Philox4x32-10; all inputs broadcastable integer arrays; returns 4 uint32 arrays."""
    c0, c1, c2, c3 = [np.asarray(c, dtype=np.uint64) & _MASK for c in np.broadcast_arrays(c0, c1, c2, c3)]
    k0, k1 = int(k0) & 0xFFFFFFFF, int(k1) & 0xFFFFFFFF
    for _ in range(10):
        p0 = _M0 * c0
        p1 = _M1 * c2
        n0 = ((p1 >> np.uint64(32)) ^ c1 ^ np.uint64(k0)) & _MASK
        n1 = p1 & _MASK
        n2 = ((p0 >> np.uint64(32)) ^ c3 ^ np.uint64(k1)) & _MASK
        n3 = p0 & _MASK
        c0, c1, c2, c3 = n0, n1, n2, n3
        k0 = (k0 + _W0) & 0xFFFFFFFF
        k1 = (k1 + _W1) & 0xFFFFFFFF
    return c0, c1, c2, c3


def draw(seed, stream, a, b, t):
    """counter = (a, b, (u32)t, stream ^ ((u32)(t>>32) << 8)), key = seed."""
    t = int(t) & 0xFFFFFFFFFFFFFFFF
    c3 = (stream ^ (((t >> 32) & 0xFFFFFFFF) << 8)) & 0xFFFFFFFF
    return philox4x32_10(a, b, t & 0xFFFFFFFF, c3, seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)


def u23(w):
    """((w >> 9) + 0.5) / 2^23, the 23-bit uniform the device uses for Box-Muller."""
    return ((np.asarray(w, dtype=np.uint64) >> np.uint64(9)).astype(np.float64) + 0.5) / 8388608.0


def gaussians3(seed, stream, atoms, replica, t):
    """Three N(0,1) per atom: Box-Muller on words (0,1) -> (n0, n1) and (2,3) -> n2 (cos branch)."""
    w0, w1, w2, w3 = draw(seed, stream, np.asarray(atoms), replica, t)
    r1 = np.sqrt(-2.0 * np.log(u23(w0)))
    r2 = np.sqrt(-2.0 * np.log(u23(w2)))
    a1 = 2.0 * np.pi * u23(w1)
    a2 = 2.0 * np.pi * u23(w3)
    return np.stack([r1 * np.cos(a1), r1 * np.sin(a1), r2 * np.cos(a2)], axis=1)


# ---------------------------------------------------------------------------------------------
# potential energy and forces
# ---------------------------------------------------------------------------------------------

class OracleSystem:
    """f64 energy/force evaluation for a system description dict (openmmtools_amd.system.system_to_desc)."""

    def __init__(self, desc):
        self.d = desc
        self.N = int(desc['n_atoms'])
        self.mass = np.asarray(desc['mass'], dtype=np.float64)
        self.constraints = self._constraint_list()

    def _constraint_list(self):
        d = self.d
        cons = []
        for (o, h1, h2) in np.asarray(d['settle_atoms']).reshape(-1, 3):
            cons += [(o, h1, d['settle_dOH']), (o, h2, d['settle_dOH']), (h1, h2, d['settle_dHH'])]
        for atoms, dist in zip(np.asarray(d['shake_atoms']).reshape(-1, 4), np.asarray(d['shake_dist']).reshape(-1, 3)):
            for k in range(1, 4):
                if atoms[k] >= 0:
                    cons.append((atoms[0], atoms[k], dist[k - 1]))
        return cons

    # each term returns (energy, forces[N,3])
    def ext(self, x):
        d = self.d
        f = np.zeros_like(x)
        e = 0.0
        if d['n_ext'] > 0:
            idx = np.asarray(d['ext_atoms'])
            K, x0, U0 = d['ext_K'], d['ext_x0'], d['ext_U0']
            dx = x[idx].copy()
            dx[:, 0] -= x0
            e = float(0.5 * K * np.sum(dx * dx) + U0 * len(idx))    # testsystems.py:779
            f[idx] = -K * dx
        return e, f

    def energy_forces(self, x, box=None, lambda_sterics=1.0, lambda_electrostatics=1.0, forces=True, classes=None):
        e, f = self.ext(x)
        if classes is not None and 0 not in classes:            # (force class 0 = the external force: remd_set_force_groups)
            e, f = 0.0, np.zeros_like(x)
        for term in getattr(self, 'extra_terms', []):
            et, ft = term(x, box, lambda_sterics, lambda_electrostatics)
            e += et
            f += ft
        return e, f

    def potential(self, x, box=None, **kw):
        return self.energy_forces(x, box, **kw)[0]


# ---------------------------------------------------------------------------------------------
# constraints: iterative SHAKE (positions) and RATTLE (velocities), f64, tolerance 1e-12
# ---------------------------------------------------------------------------------------------

def _color_constraints(cons):
    """Group constraints into colours whose members share no atom (vectorised Gauss-Seidel sweeps)."""
    colours = []
    for c in cons:
        for col in colours:
            if c[0] not in col['atoms'] and c[1] not in col['atoms']:
                col['atoms'].update((c[0], c[1]))
                col['list'].append(c)
                break
        else:
            colours.append({'atoms': {c[0], c[1]}, 'list': [c]})
    out = []
    for col in colours:
        a = np.array(col['list'], dtype=np.float64)
        out.append((a[:, 0].astype(int), a[:, 1].astype(int), a[:, 2]))
    return out


_colour_cache = {}


def _colours(cons):
    key = id(cons)
    if key not in _colour_cache:
        _colour_cache[key] = _color_constraints(cons)
    return _colour_cache[key]


def shake(cons, invm, x_old, x_new, tol=1e-12, max_iter=500):
    """Iterative SHAKE: move x_new along the OLD bond vectors until every |r|^2 matches d^2 (relative tol).
    Constraints of one colour share no atoms and are updated together (plain Gauss-Seidel otherwise)."""
    x = x_new.copy()
    cols = _colours(cons)
    for _ in range(max_iter):
        worst = 0.0
        for (i, j, dist) in cols:
            r = x[j] - x[i]
            diff = dist * dist - np.einsum('ij,ij->i', r, r)
            worst = max(worst, float(np.max(np.abs(diff) / (dist * dist))))
            r0 = x_old[j] - x_old[i]
            lam = diff / (2.0 * (invm[i] + invm[j]) * np.einsum('ij,ij->i', r, r0))
            x[i] -= (lam * invm[i])[:, None] * r0
            x[j] += (lam * invm[j])[:, None] * r0
        if worst < tol:
            break
    return x


def rattle(cons, invm, x, v, tol=1e-12, max_iter=500):
    """Iterative RATTLE velocity stage: remove the velocity components along the constraints."""
    v = v.copy()
    cols = _colours(cons)
    for _ in range(max_iter):
        worst = 0.0
        for (i, j, dist) in cols:
            r = x[j] - x[i]
            rv = np.einsum('ij,ij->i', r, v[j] - v[i])
            worst = max(worst, float(np.max(np.abs(rv) / (dist * dist))))
            lam = rv / (np.einsum('ij,ij->i', r, r) * (invm[i] + invm[j]))
            v[i] += (lam * invm[i])[:, None] * r
            v[j] -= (lam * invm[j])[:, None] * r
        if worst < tol:
            break
    return v


# ---------------------------------------------------------------------------------------------
# Langevin splitting integrator (integrators.py:1309-1317, 1404-1460)
# ---------------------------------------------------------------------------------------------

class OracleLangevin:
    def __init__(self, system, splitting, timestep, collision_rate, n_steps, seed, cmm_frequency=None):
        self.s = system
        self.tokens = splitting.upper().split()
        self.nV = sum(t[0] == 'V' for t in self.tokens)
        # integrators.py:1507-1535: more than one distinct force group named => multiple-time-step: V<g> kicks with the forces of
        # group g and dt / (number of V<g>); otherwise every V uses all forces and dt / (number of V)
        groups = sorted(set(t[1:] for t in self.tokens if t[0] == 'V' and len(t) > 1))
        self.mts = len(groups) > 1
        self.nVg = {g: sum(t == 'V' + g for t in self.tokens) for g in groups} if self.mts else {}
        fg = np.asarray(system.d.get('force_groups', np.zeros(6, dtype=np.int32))) if hasattr(system, 'd') else np.zeros(6, dtype=np.int32)
        self.group_classes = {g: set(c for c in range(6) if int(fg[c]) == int(g)) for g in groups}
        self.nR = self.tokens.count('R')
        self.nO = self.tokens.count('O')
        self.dt, self.gamma, self.n_steps, self.seed = float(timestep), float(collision_rate), int(n_steps), int(seed)
        h = self.dt / max(1, self.nO)
        self.a = math.exp(-self.gamma * h)                         # integrators.py:1143
        self.b = math.sqrt(1.0 - math.exp(-2.0 * self.gamma * h))  # :1146
        self.cmm = system.d['cmm_frequency'] if cmm_frequency is None else cmm_frequency

    def assign_velocities(self, x, kT, replica, iteration):
        """mcmc.py:710-711: Maxwell-Boltzmann draw, then velocity constraints."""
        s = self.s
        xi = gaussians3(self.seed, STREAM_VELOCITY, np.arange(s.N), replica, iteration)
        v = np.sqrt(kT / s.mass)[:, None] * xi
        if s.constraints:
            v = rattle(s.constraints, 1.0 / s.mass, x, v)
        return v

    def run(self, x, v, box, kT, replica, iteration, first_step=0, n_steps=None, tokens=None,
            lambda_sterics=1.0, lambda_electrostatics=1.0, barostat=None):
        """barostat: None or dict(obj=OracleBarostat, pressure=p, frequency=f, steps_done=n, attempts_done=m); when given the
        (possibly rescaled) box is returned as a third value."""
        s = self.s
        invm = 1.0 / s.mass
        tokens = self.tokens if tokens is None else tokens
        n_steps = self.n_steps if n_steps is None else n_steps
        x, v = x.copy(), v.copy()
        f = None
        work = getattr(self, 'work', None)            # dict(heat, shadow_work, n_accepted, n_trials) or None: measured when present
        ke = lambda vv: kinetic_energy(s.mass, vv)
        pe = lambda xx: s.energy_forces(xx, box, lambda_sterics, lambda_electrostatics)[0]
        xold = vold = None
        baro_steps = barostat['steps_done'] if barostat else 0
        baro_attempt = barostat['attempts_done'] if barostat else 0
        for step in range(n_steps):
            gstep = iteration * self.n_steps + first_step + step
            if self.cmm and ((first_step + step) % self.cmm) == 0:
                p = (s.mass[:, None] * v).sum(axis=0)                # CMMotionRemover at the top of a step
                v -= p / s.mass.sum()
            if barostat:
                baro_steps += 1
                if baro_steps % barostat['frequency'] == 0:          # MonteCarloBarostatImpl: every frequency-th step
                    x, box, _ = barostat['obj'].attempt(x, box, kT, barostat['pressure'], replica, baro_attempt,
                                                        long_range=barostat.get('long_range', 0.0),
                                                        lambda_sterics=lambda_sterics,
                                                        lambda_electrostatics=lambda_electrostatics)
                    baro_attempt += 1
                    f = None
            oidx = 0
            brace = 0
            for tok in tokens:
                if tok == '{':                                                      # integrators.py:1539-1542
                    xold, vold = x.copy(), v.copy()
                elif tok == '}':                                                    # :1544-1557
                    w = draw(self.seed, STREAM_METROPOLIS, brace, replica, gstep)
                    u = ((int(w[2]) << 21) | (int(w[3]) >> 11)) / 9007199254740992.0
                    work['n_trials'] += 1
                    if np.exp(-work['shadow_work'] / kT) - u >= 0.0:
                        work['n_accepted'] += 1
                    else:
                        x, v = xold.copy(), -vold
                        f = None
                    work['shadow_work'] = 0.0
                    brace += 1
                elif tok[0] == 'V' and self.mts:
                    g = tok[1:]
                    fgrp = s.energy_forces(x, box, lambda_sterics, lambda_electrostatics, classes=self.group_classes[g])[1]
                    ke0 = ke(v) if work is not None else 0.0
                    v = v + (self.dt / self.nVg[g]) * fgrp * invm[:, None]          # :1437-1438
                    if s.constraints:
                        v = rattle(s.constraints, invm, x, v)
                    if work is not None:
                        work['shadow_work'] += ke(v) - ke0
                elif tok[0] == 'V':
                    if f is None:
                        f = s.energy_forces(x, box, lambda_sterics, lambda_electrostatics)[1]
                    ke0 = ke(v) if work is not None else 0.0
                    v = v + (self.dt / self.nV) * f * invm[:, None]                 # :1440-1442
                    if s.constraints:
                        v = rattle(s.constraints, invm, x, v)
                    if work is not None:
                        work['shadow_work'] += ke(v) - ke0                          # :1444-1446
                elif tok == 'R':
                    if work is not None:
                        e0 = ke(v) + pe(x)                                          # :1407-1409
                    h = self.dt / self.nR
                    x1 = x + h * v                                                  # :1414
                    if s.constraints:
                        xc = shake(s.constraints, invm, x, x1)                      # :1416
                        v = v + (xc - x1) / h                                       # :1417
                        x = xc
                        v = rattle(s.constraints, invm, x, v)                       # :1418
                    else:
                        x = x1
                    f = None
                    if work is not None:
                        work['shadow_work'] += ke(v) + pe(x) - e0                   # :1420-1423
                elif tok == 'O':
                    ke0 = ke(v) if work is not None else 0.0
                    cnt = gstep * max(1, self.nO) + oidx
                    xi = gaussians3(self.seed, STREAM_OU, np.arange(s.N), replica, cnt)
                    v = self.a * v + self.b * np.sqrt(kT * invm)[:, None] * xi      # :1455
                    if s.constraints:
                        v = rattle(s.constraints, invm, x, v)
                    oidx += 1
                    if work is not None:
                        work['heat'] += ke(v) - ke0                                 # :1457-1460
        if barostat:
            return x, v, box
        return x, v


def kinetic_energy(mass, v):
    return float(0.5 * np.sum(mass[:, None] * v * v))


def reduced_potential_matrix(potentials, betas, energy_const=None, alch=None):
    """u[r, l] = beta_l (U_r + const_l + alch[r, l])  (states.py:1908-1917, paralleltempering.py:206-215)."""
    U = np.asarray(potentials, dtype=np.float64)[:, None]
    tot = U + (0.0 if energy_const is None else np.asarray(energy_const)[None, :])
    if alch is not None:
        tot = tot + alch
    return np.asarray(betas)[None, :] * tot


class OracleFIRE:
    """f64 restatement of the reference's FIREMinimizationIntegrator (openmmtools/integrators.py:2290-2469), one replica at
    a time, as MultiStateSampler._minimize_replica drives it (multistatesampler.py:1351-1434).  TEST INFRASTRUCTURE ONLY."""

    def __init__(self, system, tolerance=0.0, timestep=0.001, alpha=0.1, dt_max=0.010, f_inc=1.1, f_dec=0.5, f_alpha=0.99,
                 n_min=5):
        self.s = system
        self.ftol, self.timestep, self.alpha0 = float(tolerance), float(timestep), float(alpha)
        self.dt_max, self.f_inc, self.f_dec, self.f_alpha, self.n_min = dt_max, f_inc, f_dec, f_alpha, int(n_min)

    def minimize(self, x, box=None, max_iterations=0, lambda_sterics=1.0, lambda_electrostatics=1.0, history=None):
        s = self.s
        invm = 1.0 / s.mass
        x = np.array(x, dtype=np.float64)
        v = np.zeros_like(x)                                         # :2341
        dt, alpha, n_neg, converged = self.timestep, self.alpha0, 0, False
        ndof = 3 * s.N
        ef = lambda y: s.energy_forces(y, box, lambda_sterics, lambda_electrostatics)
        E, f = ef(x)
        it = 0
        limit = max_iterations if max_iterations > 0 else 200000
        while it < limit:
            if np.sqrt((f * f).sum()) / ndof <= self.ftol:           # :2377-2386
                converged = True
            if converged:
                if max_iterations == 0:
                    break
                it += 1
                continue
            x0, v0, E0, f0 = x, v, E, f                              # :2392-2394
            v = v + 0.5 * dt * f * invm[:, None]                     # :2397
            x1 = x + dt * v                                          # :2398-2399
            xn = shake(s.constraints, invm, x, x1) if s.constraints else x1      # :2400
            En, fn = ef(xn)
            v = v + 0.5 * dt * fn * invm[:, None] + (xn - x1) / dt   # :2401
            if s.constraints:
                v = rattle(s.constraints, invm, xn, v)               # :2402
            dE = En - E0                                             # :2404
            fmag, vmag = np.sqrt((fn * fn).sum()), np.sqrt((v * v).sum())        # :2408-2413
            P = float((fn * v).sum())                                # :2416
            if fmag > 0:
                v = (1.0 - alpha) * v + alpha * (fn / fmag) * vmag   # :2421
            x, E, f = xn, En, fn
            if not (dE < 0):                                         # :2423-2431
                x, v, E, f, P = x0, v0, E0, f0, -1.0
            if dt <= 1.0e-5 * self.timestep:                         # :2433-2437
                converged = True
            if P > 0:                                                # :2439-2449
                n_neg += 1
                if n_neg > self.n_min:
                    dt = min(dt * self.f_inc, self.dt_max)
                    alpha *= self.f_alpha
            if P < 0:                                                # :2451-2458
                n_neg = 0
                dt *= self.f_dec
                v = np.zeros_like(v)
                alpha = self.alpha0
            it += 1
            if history is not None:
                history.append((E, dt, alpha, n_neg))
        return x, v, E, converged, it


# ---------------------------------------------------------------------------------------------
# Monte Carlo barostat: f64 restatement of OpenMM's MonteCarloBarostatImpl::updateContextState (the NPT machinery the
# reference relies on, openmmtools/states.py:1177-1181; integrators.py:1313).  TEST INFRASTRUCTURE ONLY.
# ---------------------------------------------------------------------------------------------
STREAM_BAROSTAT = 6


def molecules_from_desc(desc):
    """Connected components over exceptions (1-2, 1-3, 1-4 ... pairs), bonds and constraints; the device uses the same
    union (forces.hip: remd_build_nonbonded groups).  Returns a list of atom-index arrays."""
    n = int(desc['n_atoms'])
    parent = list(range(n))

    def find(a):
        while parent[a] != a:
            parent[a] = parent[parent[a]]
            a = parent[a]
        return a

    def union(a, b):
        ra, rb = find(int(a)), find(int(b))
        if ra != rb:
            parent[max(ra, rb)] = min(ra, rb)
    for key, width in (('exception_atoms', 2), ('bond_atoms', 2)):
        arr = np.asarray(desc.get(key, []), dtype=int).reshape(-1, width)
        for a, b in arr:
            union(a, b)
    for (o, h1, h2) in np.asarray(desc['settle_atoms'], dtype=int).reshape(-1, 3):
        union(o, h1); union(o, h2)
    for atoms in np.asarray(desc['shake_atoms'], dtype=int).reshape(-1, 4):
        for k in range(1, 4):
            if atoms[k] >= 0:
                union(atoms[0], atoms[k])
    groups = {}
    for a in range(n):
        groups.setdefault(find(a), []).append(a)
    return [np.array(g) for g in groups.values()]


def u53(hi, lo):
    return float(((int(hi) << 21) | (int(lo) >> 11)) / 9007199254740992.0)


class OracleBarostat:
    def __init__(self, system, seed, molecules):
        self.s, self.seed, self.mols = system, int(seed), molecules
        self.state = {}          # replica -> [volume_scale, attempted, accepted, total_attempted, total_accepted]

    def attempt(self, x, box, kT, pressure, replica, attempt, long_range=0.0, **lam):
        """long_range: coefficient c of a state constant c / V that the potential does not carry (alchemical sterics LRC)."""
        st = self.state.setdefault(replica, [0.0, 0, 0, 0, 0])
        box = np.asarray(box, dtype=np.float64)
        U0 = self.s.potential(x, box, **lam)
        V = float(np.prod(box))
        if st[0] <= 0.0:
            st[0] = 0.01 * V
        w = draw(self.seed, STREAM_BAROSTAT, 0, replica, attempt)
        dV = st[0] * 2.0 * (u53(w[2], w[3]) - 0.5)
        newV = V + dV
        scale = (newV / V) ** (1.0 / 3.0)
        xn = x.copy()
        for m in self.mols:
            c = x[m].mean(axis=0)
            cw = c - np.floor(c / box) * box
            xn[m] += cw * (scale - 1.0) - (c - cw)
        boxn = box * scale
        U1 = self.s.potential(xn, boxn, **lam)
        wgt = U1 - U0 + long_range * (1.0 / newV - 1.0 / V) + pressure * dV - len(self.mols) * kT * np.log(newV / V)
        q = draw(self.seed, STREAM_BAROSTAT, 1, replica, attempt)
        reject = (not (wgt <= 0.0)) and (not (u53(q[2], q[3]) <= np.exp(-wgt / kT)))
        if reject:
            xn, boxn = x, box
        else:
            st[2] += 1; st[4] += 1
        st[1] += 1; st[3] += 1
        if st[1] >= 10:
            Vc = float(np.prod(boxn))
            if st[2] < 0.25 * st[1]:
                st[0] /= 1.1; st[1] = 0; st[2] = 0
            elif st[2] > 0.75 * st[1]:
                st[0] = min(st[0] * 1.1, Vc * 0.3); st[1] = 0; st[2] = 0
        return xn, boxn, (not reject)

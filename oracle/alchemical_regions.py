"""TEST INFRASTRUCTURE (oracle): f64 restatement of the custom forces the reference's AbsoluteAlchemicalFactory builds for general
alchemical regions -- /root/reference/openmmtools/alchemy/alchemy.py:1539-2038 (_alchemically_modify_NonbondedForce) with the energy
expressions of :1356-1390 (sterics), :1392-1471 (electrostatics), :1473-1508 (reaction field), :1510-1537 (Ewald direct space).
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this; the product path (csrc/alch_regions.hip) never does.

Pinned by tests/test_alchemical_regions.py against tests/golden/reference_alchemy_expressions.json: values of the reference's OWN
expression strings (taken out of its syntax tree by tests/golden/make_golden_alchemy_strings.py) on a grid of distances, charges,
soft-core constants and lambdas.

``terms`` = the dict openmmtools_amd.alchemy.AbsoluteAlchemicalFactory._region_terms returns (system_to_desc(...)['alch_regions']):
    region_of_atom [N] (0 = environment), softcore [n][8] = alpha, beta, a, b, c, d, e, f, annihilate [n][2] = sterics, electrostatics,
    interactions [m][2] (1-based), charge / sigma / epsilon [N] of the REFERENCE NonbondedForce, the exceptions that became custom bonds,
    electrostatics 0 / 1, elec_alpha, elec_krf, elec_crf, elec_switch_distance.
Total potential of an alchemical System in that mode = ForceFieldOracle(descriptor of the factory's NonbondedForce + bonded forces)
+ RegionOracle (this file).
"""
import numpy as np
import torch

ONE_4PI_EPS0 = 138.93545764438198


class RegionOracle:
    def __init__(self, terms, cutoff, switch_distance, exclusions):
        """cutoff / switch_distance (< 0 or None: no switch): the NonbondedForce's; exclusions: every exception pair of the System
        (the custom nonbonded forces exclude them all, alchemy.py:1944-1947)."""
        t = terms
        self.terms = terms
        self.g = np.asarray(t['region_of_atom'], dtype=int)
        self.N = len(self.g)
        self.sc = np.asarray(t['softcore'], dtype=np.float64).reshape(-1, 8)
        self.ann = np.asarray(t['annihilate'], dtype=int).reshape(-1, 2)
        self.n = len(self.sc)
        self.inter = [(int(a), int(b)) for a, b in np.asarray(t['interactions'], dtype=int).reshape(-1, 2)]
        self.q = np.asarray(t['charge'], dtype=np.float64)
        self.sig = np.asarray(t['sigma'], dtype=np.float64)
        self.eps = np.asarray(t['epsilon'], dtype=np.float64)
        self.exc_atoms = np.asarray(t['exception_atoms'], dtype=int).reshape(-1, 2)
        self.exc_params = np.asarray(t['exception_params'], dtype=np.float64).reshape(-1, 3)
        self.elec = bool(t['electrostatics'])
        self.alpha, self.krf, self.crf = float(t['elec_alpha']), float(t['elec_krf']), float(t['elec_crf'])
        self.rs_e = float(t['elec_switch_distance'])
        self.rc = float(cutoff) if cutoff and cutoff > 0 else float('inf')           # (0: a NoCutoff system)
        self.rs = -1.0 if switch_distance is None else float(switch_distance)
        excl = set((min(int(i), int(j)), max(int(i), int(j))) for i, j in exclusions)
        # the candidate pairs: (alchemical, environment) and alchemical pairs once, of regions that interact, not excluded
        alch = np.nonzero(self.g)[0]
        pairs, kinds = [], []
        for a in alch:
            for j in range(self.N):
                if j == a or (self.g[j] > 0 and j < a) or (min(a, j), max(a, j)) in excl:
                    continue
                k = self._class(self.g[a], self.g[j])
                if k is None:
                    continue
                pairs.append((a, j)); kinds.append(k)
        self.pairs = np.array(pairs, dtype=int).reshape(-1, 2)
        self.kinds = kinds
        # (an exception between atoms of two regions is a bond of the FIRST region's (environment, region) force: the factory's loop
        # meets it there as "only one alchemical" and zeroes it before the second region's turn, alchemy.py:1972-1976, 1992-2006)
        self.exc_kinds = [self._class(0, min(self.g[i], self.g[j])) if (self.g[i] and self.g[j] and self.g[i] != self.g[j])
                          else self._class(self.g[i], self.g[j]) for i, j in self.exc_atoms]

    def _class(self, ga, gb):
        """(kind, a, b, P): kind 0 (environment, a), 1 (a, a), 2 (a, b) interacting; P = region of the soft-core constants (1-based)"""
        if ga == 0 or gb == 0:
            y = max(ga, gb)
            return (0, y, y, y)
        if ga == gb:
            return (1, ga, ga, ga)
        for a, b in self.inter:
            if {a, b} == {ga, gb}:
                return (2, a, b, b)
        return None

    def _lambdas(self, cls, ls, le):
        kind, a, b, _ = cls
        if kind == 0:
            return ls[a - 1], le[a - 1]
        if kind == 1:
            return (ls[a - 1] if self.ann[a - 1, 0] else 1.0), (le[a - 1] if self.ann[a - 1, 1] else 1.0)
        return ls[a - 1] * ls[b - 1], le[a - 1] * le[b - 1]

    @staticmethod
    def _switch(r, rs, rc):
        if rs is None or rs < 0 or rs >= rc:
            return torch.ones_like(r)
        x = torch.clamp((r - rs) / (rc - rs), 0.0, 1.0)
        return 1.0 - 10.0 * x ** 3 + 15.0 * x ** 4 - 6.0 * x ** 5

    def _sterics(self, r, sigma, eps, l, P):
        alpha, _, a, b, c = self.sc[P - 1][:5]
        reff = sigma * (alpha * (1.0 - l) ** b + (r / sigma) ** c) ** (1.0 / c)                 # alchemy.py:1388
        x = (sigma / reff) ** 6
        return (l ** a) * 4.0 * eps * x * (x - 1.0)                                              # :1385-1386

    def _electrostatics(self, r, sigma, qq, l, P, alpha, krf, crf):
        beta, d, e, f = self.sc[P - 1][1], self.sc[P - 1][5], self.sc[P - 1][6], self.sc[P - 1][7]
        reff = sigma * (beta * (1.0 - l) ** e + (r / sigma) ** f) ** (1.0 / f)                  # :1429
        g = (torch.erfc(alpha * reff) if alpha > 0 else torch.ones_like(reff)) / reff + krf * reff ** 2 - crf      # :1434, 1505-1507, 1534-1536
        return (l ** d) * ONE_4PI_EPS0 * qq * g                                                  # :1425-1426

    def energy_torch(self, x, box, ls, le):
        """x: torch [N][3]; box: edge lengths or None; ls / le: lambda_sterics / lambda_electrostatics per region."""
        e = x.new_zeros(())
        box_t = None if box is None else torch.tensor(np.asarray(box, dtype=np.float64))
        if len(self.pairs):
            i, j = self.pairs[:, 0], self.pairs[:, 1]
            d = x[j] - x[i]
            if box_t is not None:
                d = d - box_t * torch.round(d / box_t)
            r = d.norm(dim=1)
            inside = r < self.rc
            sigma = torch.tensor(0.5 * (self.sig[i] + self.sig[j]))
            eps = torch.tensor(np.sqrt(self.eps[i] * self.eps[j]))
            qq = torch.tensor(self.q[i] * self.q[j])
            for cls in sorted(set(self.kinds)):
                m = torch.tensor([k == cls for k in self.kinds]) & inside
                if not bool(m.any()):
                    continue
                l_s, l_e = self._lambdas(cls, ls, le)
                rr = r[m]
                u = self._sterics(rr, sigma[m], eps[m], l_s, cls[3]) * self._switch(rr, self.rs, self.rc)
                e = e + torch.where(eps[m] != 0, u, torch.zeros_like(u)).sum()
                if self.elec:
                    u = self._electrostatics(rr, sigma[m], qq[m], l_e, cls[3], self.alpha, self.krf, self.crf) * self._switch(rr, self.rs_e, self.rc)
                    e = e + torch.where(qq[m] != 0, u, torch.zeros_like(u)).sum()
        for t, (i, j) in enumerate(self.exc_atoms):
            qq, sg, ep = self.exc_params[t]
            d = x[j] - x[i]
            if box_t is not None:
                d = d - box_t * torch.round(d / box_t)
            r = d.norm().reshape(1)
            l_s, l_e = self._lambdas(self.exc_kinds[t], ls, le)
            sg_t = torch.tensor([sg])
            if ep != 0.0:
                e = e + self._sterics(r, sg_t, torch.tensor([ep]), l_s, self.exc_kinds[t][3]).sum()
            if self.elec and qq != 0.0:
                a_, k_, c_ = (self.alpha, self.krf, self.crf) if self.terms.get('consistent_exceptions', 0) else (0.0, 0.0, 0.0)
                e = e + self._electrostatics(r, sg_t, torch.tensor([qq]), l_e, self.exc_kinds[t][3], a_, k_, c_).sum()      # :1434, 1456-1461
        return e

    def bonded_torch(self, x, lb, la, lt):
        """softened bonded terms: lambda_{bonds, angles, torsions} of the term's region x the reference's harmonic bond / harmonic angle /
        periodic torsion (alchemy.py:1341 'lambda_bonds*(K/2)*(r-r0)^2', :1261 'lambda_angles*(K/2)*(theta-theta0)^2',
        :1180 'lambda_torsions*k*(1+cos(periodicity*theta-phase))'); lb / la / lt: per region, None = 1"""
        t = self.terms
        e = x.new_zeros(())
        lam = lambda v, reg: torch.ones(len(reg), dtype=torch.float64) if v is None else torch.tensor(np.asarray(v, dtype=np.float64)[np.asarray(reg, dtype=int) - 1])
        ba = np.asarray(t.get('bond_atoms', np.zeros((0, 2))), dtype=int).reshape(-1, 2)
        if len(ba):
            bp = torch.tensor(np.asarray(t['bond_params'], dtype=np.float64).reshape(-1, 2))
            r = (x[ba[:, 1]] - x[ba[:, 0]]).norm(dim=1)
            e = e + (lam(lb, t['bond_region']) * 0.5 * bp[:, 1] * (r - bp[:, 0]) ** 2).sum()
        aa = np.asarray(t.get('angle_atoms', np.zeros((0, 3))), dtype=int).reshape(-1, 3)
        if len(aa):
            ap = torch.tensor(np.asarray(t['angle_params'], dtype=np.float64).reshape(-1, 2))
            v0, v1 = x[aa[:, 0]] - x[aa[:, 1]], x[aa[:, 2]] - x[aa[:, 1]]
            th = torch.acos(torch.clamp((v0 * v1).sum(1) / (v0.norm(dim=1) * v1.norm(dim=1)), -1.0, 1.0))
            e = e + (lam(la, t['angle_region']) * 0.5 * ap[:, 1] * (th - ap[:, 0]) ** 2).sum()
        ta = np.asarray(t.get('torsion_atoms', np.zeros((0, 4))), dtype=int).reshape(-1, 4)
        if len(ta):
            tp = torch.tensor(np.asarray(t['torsion_params'], dtype=np.float64).reshape(-1, 3))
            b1, b2, b3 = x[ta[:, 1]] - x[ta[:, 0]], x[ta[:, 2]] - x[ta[:, 1]], x[ta[:, 3]] - x[ta[:, 2]]
            m, n = torch.linalg.cross(b1, b2), torch.linalg.cross(b2, b3)
            phi = torch.atan2(b2.norm(dim=1) * (b1 * n).sum(1), (m * n).sum(1))
            e = e + (lam(lt, t['torsion_region']) * tp[:, 2] * (1.0 + torch.cos(tp[:, 0] * phi - tp[:, 1]))).sum()
        return e

    def energy_forces(self, x, box, ls, le, forces=True, bonded=(None, None, None)):
        xt = torch.tensor(np.asarray(x, dtype=np.float64), requires_grad=forces)
        e = self.energy_torch(xt, box, ls, le) + self.bonded_torch(xt, *bonded)
        if not forces or not e.requires_grad:
            return float(e.detach()), np.zeros((self.N, 3))
        (g,) = torch.autograd.grad(e, xt)
        return float(e.detach()), -g.numpy()

    def state_energies(self, x, box, LS, LE):
        """the region terms at every state's lambdas: LS / LE [K][n]"""
        xt = torch.tensor(np.asarray(x, dtype=np.float64))
        return np.array([float(self.energy_torch(xt, box, ls, le)) for ls, le in zip(LS, LE)])


class _Total:
    """factory's NonbondedForce + bonded terms (ForceFieldOracle) + custom forces (RegionOracle); under the exact PME treatment
    (terms['exact_pme'], alchemy.py:1663-1681, 1893-1899, 1978-1982) the NonbondedForce is evaluated WITH its parameter offsets: the
    reference charges of the alchemical atoms times their region's lambda_electrostatics inside the whole Ewald sum, the charge products
    of the exceptions that touch a region times the (first) region's lambda, every pair of atoms of two regions that do not interact
    excluded (:1663-1672); the custom forces are then sterics only (of two interacting regions: none, :1886-1911)."""

    def __init__(self, desc):
        from .forcefield import ForceFieldOracle
        terms = desc['alch_regions']
        self.exact = bool(terms.get('exact_pme', 0))
        excl = np.asarray(desc['exception_atoms']).reshape(-1, 2)
        rs = desc['switch_distance'] if desc['switch_distance'] > 0 else None
        if not self.exact:
            self.base = ForceFieldOracle(desc)
            self.reg = RegionOracle(terms, desc['cutoff'], rs, excl)
            return
        g = np.asarray(terms['region_of_atom'], dtype=int)
        self.g = g
        inter = set(frozenset((int(a), int(b))) for a, b in np.asarray(terms['interactions'], dtype=int).reshape(-1, 2))
        d = dict(desc)
        d['charge'] = np.where(g > 0, np.asarray(terms['charge'], dtype=np.float64), np.asarray(desc['charge'], dtype=np.float64))
        exc_a = [tuple(int(v) for v in e) for e in excl]
        exc_p = np.array(desc['exception_params'], dtype=np.float64).reshape(-1, 3).copy()
        took = {frozenset((int(i), int(j))): p for (i, j), p in zip(np.asarray(terms['exception_atoms']).reshape(-1, 2), np.asarray(terms['exception_params']).reshape(-1, 3))}
        self.exc_region = np.zeros(len(exc_a) , dtype=int)
        for k, (i, j) in enumerate(exc_a):
            if frozenset((i, j)) in took:
                exc_p[k, 0] = took[frozenset((i, j))][0]               # the charge product comes back as an offset; the LJ part stays with the custom bonds
                self.exc_region[k] = min(v for v in (g[i], g[j]) if v > 0)
        have = set(frozenset(e) for e in exc_a)
        extra = [(int(i), int(j)) for i in np.nonzero(g)[0] for j in np.nonzero(g)[0]
                 if i < j and g[i] != g[j] and frozenset((int(g[i]), int(g[j]))) not in inter and frozenset((int(i), int(j))) not in have]
        d['exception_atoms'] = np.array(exc_a + extra, dtype=np.int32).reshape(-1, 2)
        d['exception_params'] = np.vstack([exc_p, np.tile([0.0, 1.0, 0.0], (len(extra), 1))]) if extra else exc_p
        self.exc_region = np.concatenate([self.exc_region, np.zeros(len(extra), dtype=int)])
        d.pop('alch_regions')
        self.base = ForceFieldOracle(d)
        self.reg = RegionOracle(dict(terms, electrostatics=0, interactions=np.zeros((0, 2), dtype=np.int32)), desc['cutoff'], rs, excl)

    def _set(self, le):
        if self.exact:
            lam = np.concatenate([[1.0], np.asarray(le, dtype=np.float64)])
            self.base.q_scale = torch.tensor(lam[self.g])
            self.base.exc_scale = torch.tensor(lam[self.exc_region])

    def energy_forces(self, x, box, ls, le, forces=True, bonded=(None, None, None)):
        self._set(le)
        e0, f0 = self.base.energy_forces(x, box, forces=forces)
        e1, f1 = self.reg.energy_forces(x, box, ls, le, forces=forces, bonded=bonded)
        return e0 + e1, (f0 + f1 if forces else None)


def total_state_energies(desc, x, box, LS, LE, BONDED=None):
    """Potential of an alchemical System in the general-regions mode at every state; BONDED: [K][3][n] lambda_bonds / angles / torsions"""
    t = _Total(desc)
    return np.array([t.energy_forces(x, box, ls, le, forces=False, bonded=(None, None, None) if BONDED is None else tuple(BONDED[k]))[0]
                     for k, (ls, le) in enumerate(zip(LS, LE))])


def total_energy_forces(desc, x, box, ls, le, bonded=(None, None, None)):
    return _Total(desc).energy_forces(x, box, ls, le, bonded=bonded)

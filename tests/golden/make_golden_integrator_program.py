"""The reference's Langevin integrator EXECUTED: its step program and the trajectory that program produces.

openmmtools/integrators.py builds an openmm.CustomIntegrator: `LangevinIntegrator.__init__` parses the splitting string and calls
addComputePerDof / addComputeSum / addComputeGlobal / addConstrain* / beginIfBlock ... with expression strings (integrators.py:1093-1160,
1404-1460, 1539-1557).  openmm is absent here, but those calls ARE the algorithm.  This script
  1. takes the class LangevinIntegrator out of the module's syntax tree UNCHANGED (and ThermostatedIntegrator's
     addComputeTemperatureDependentConstants, integrators.py:235-253) and executes it on a stand-in base class that records every
     CustomIntegrator call -- the result is the literal step program the reference would hand to OpenMM, per splitting string;
  2. interprets that program the way OpenMM's CustomIntegrator does (per-degree-of-freedom expressions, sums, globals, if-blocks,
     `gaussian` / `uniform` draws, `energy`, forces f and f<group> recomputed when the positions changed) on a four-atom chain without
     constraints (bonds in force group 0, angles in group 1), for a few steps from a fixed start.
The forces are the analytic harmonic bond / angle forces written out below (checked against finite differences when this script runs;
nothing of this repository's force code is used), the normal deviates and the Metropolis uniforms are the ones this repository's engines draw for (seed, replica 0, step, index
of the O / '}' in the program) so that the engines can be run on the same noise -- they are stored in the fixture.
Output: tests/golden/integrator_program_reference.json (programs as text, start state, noise, x and v after every step, heat / shadow
work / acceptance counters).  tests/test_integrator_program.py runs the oracle integrator and the C++ port against it.

usage: python tests/golden/make_golden_integrator_program.py         (/root/reference is not needed by the tests)"""
import ast
import json
import os
import re
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import md_oracle                                   # noqa: E402  (the engines' noise streams only)

SRC = '/root/reference/openmmtools/integrators.py'
OUT = os.path.join(HERE, 'integrator_program_reference.json')
KB = 0.008314462618153242          # kJ/mol/K (openmmtools/constants.py kB in MD units)
SEED, REPLICA, N_STEPS = 0x5EED1234, 0, 6
TEMPERATURE, GAMMA, DT = 310.0, 5.0, 0.001                      # K, 1/ps, ps

CASES = [  # (splitting, measure_shadow_work, measure_heat[, timestep])
    ('V R O R V', False, False), ('O V R V O', False, False), ('V R R O R R V', False, False), ('R V O V R', False, False),
    ('V R O R V', True, True), ('O { V R V } O', False, True), ('V0 V1 R O R V1 V0', False, False), ('O V R V O R O', False, True),
    ('O { V R V } O', False, False, 0.003), ('V1 V0 R R O R R V0 V1', True, False, 0.002),
]


class unit:
    kelvin = picoseconds = 1.0
    femtoseconds = 1.0e-3


class Recorder:
    """what the reference's class calls on openmm.CustomIntegrator / ThermostatedIntegrator, recorded"""
    def __init__(self, temperature, timestep):
        self.program, self.globals, self.per_dof = [], {'kT': KB * temperature}, []
        self.dt = float(timestep)
    global_variable_names = property(lambda self: list(self.globals))
    def addGlobalVariable(self, name, value): self.globals[name] = float(value)
    def addPerDofVariable(self, name, value): self.per_dof.append(name)
    def setConstraintTolerance(self, tol): self.constraint_tolerance = tol
    def addUpdateContextState(self): self.program.append(['update_context_state'])
    def addComputePerDof(self, var, expr): self.program.append(['per_dof', var, expr])
    def addComputeGlobal(self, var, expr): self.program.append(['global', var, expr])
    def addComputeSum(self, var, expr): self.program.append(['sum', var, expr])
    def addConstrainPositions(self): self.program.append(['constrain_positions'])
    def addConstrainVelocities(self): self.program.append(['constrain_velocities'])
    def beginIfBlock(self, cond): self.program.append(['if', cond])
    def endBlock(self): self.program.append(['end'])


class PlainRecorder(Recorder):
    """openmm.CustomIntegrator(timestep): what VelocityVerletIntegrator derives from"""
    def __init__(self, timestep):
        self.program, self.globals, self.per_dof = [], {}, []
        self.dt = float(timestep)


class mm:
    CustomIntegrator = PlainRecorder


def other_reference_classes(names):
    """further integrator classes of the module, executed the same way (their bases: ThermostatedIntegrator or mm.CustomIntegrator)"""
    tree = ast.parse(open(SRC).read())
    ns = dict(np=np, numpy=np, re=re, unit=unit, logger=None, mm=mm, openmm=mm, ThermostatedIntegrator=Recorder)
    out = {}
    for name in names:
        cls = next(n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == name)
        exec(compile(ast.Module(body=[cls], type_ignores=[]), SRC, 'exec'), ns)
        out[name] = (ns[name], (cls.lineno, cls.end_lineno))
    return out


def reference_class():
    tree = ast.parse(open(SRC).read())
    thermo = next(n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == 'ThermostatedIntegrator')
    graft = next(n for n in thermo.body if isinstance(n, ast.FunctionDef) and n.name == 'addComputeTemperatureDependentConstants')
    ns = dict(np=np, numpy=np, re=re, unit=unit, logger=None)
    exec(compile(ast.Module(body=[graft], type_ignores=[]), SRC, 'exec'), ns)
    Recorder.addComputeTemperatureDependentConstants = ns['addComputeTemperatureDependentConstants']
    ns['ThermostatedIntegrator'] = Recorder
    cls = next(n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == 'LangevinIntegrator')
    exec(compile(ast.Module(body=[cls], type_ignores=[]), SRC, 'exec'), ns)
    return ns['LangevinIntegrator'], (cls.lineno, cls.end_lineno)


MASSES = (12.011, 1.008, 15.999, 14.007)
BONDS = ((0, 1, 0.109, 284512.0), (1, 2, 0.141, 267776.0), (2, 3, 0.133, 410032.0))          # i, j, r0 (nm), k (kJ/mol/nm^2): force group 0
ANGLES = ((0, 1, 2, 1.911, 418.4), (1, 2, 3, 2.094, 585.76))                                   # i, j, k, theta0 (rad), k (kJ/mol/rad^2): group 1


class Chain:
    """E = sum 1/2 k (r - r0)^2 + sum 1/2 k (theta - theta0)^2 (OpenMM's HarmonicBondForce / HarmonicAngleForce), analytic forces"""
    def energy_forces(self, x, groups=None):
        e, f = 0.0, np.zeros_like(x)
        if groups is None or 0 in groups:
            for i, j, r0, k in BONDS:
                d = x[i] - x[j]
                r = np.sqrt(d @ d)
                e += 0.5 * k * (r - r0) ** 2
                g = k * (r - r0) * d / r
                f[i] -= g; f[j] += g
        if groups is None or 1 in groups:
            for i, j, l, t0, k in ANGLES:
                u, w = x[i] - x[j], x[l] - x[j]
                nu, nw = np.sqrt(u @ u), np.sqrt(w @ w)
                c = (u @ w) / (nu * nw)
                t = np.arccos(c)
                e += 0.5 * k * (t - t0) ** 2
                dedt = k * (t - t0)
                s_ = np.sqrt(1.0 - c * c)
                dti = -(w / (nu * nw) - c * u / (nu * nu)) / s_            # d theta / d x_i
                dtl = -(u / (nu * nw) - c * w / (nw * nw)) / s_
                f[i] -= dedt * dti; f[l] -= dedt * dtl; f[j] += dedt * (dti + dtl)
        return e, f


def _check_forces(chain, x):
    for groups in (None, {0}, {1}):
        e0, f = chain.energy_forces(x, groups)
        h = 1e-6
        for a in range(x.shape[0]):
            for c in range(3):
                xp, xm = x.copy(), x.copy()
                xp[a, c] += h; xm[a, c] -= h
                fd = -(chain.energy_forces(xp, groups)[0] - chain.energy_forces(xm, groups)[0]) / (2 * h)
                assert abs(fd - f[a, c]) < 1e-5 * max(1.0, abs(fd)), (groups, a, c, fd, f[a, c])


_EQ = re.compile(r'(?<![!<>=])=(?!=)')


def _expr(text):
    return text.replace('^', '**')


def interpret(rec, chain, x, v, n_steps, noise, uniforms, keep=('heat', 'shadow_work', 'naccept', 'ntrials', 'nreject')):
    """OpenMM's CustomIntegrator semantics for the recorded program; returns per-step (x, v) and the globals at the end of each step"""
    N = x.shape[0]
    m = np.asarray(MASSES, dtype=np.float64)[:, None]
    g = dict(rec.globals); g['dt'] = rec.dt
    per = {name: np.zeros((N, 3)) for name in rec.per_dof}
    funcs = dict(sqrt=np.sqrt, exp=np.exp, step=lambda t: (np.asarray(t) >= 0) * 1.0)
    x, v = x.copy(), v.copy()
    out = []
    for step in range(n_steps):
        gauss = list(noise[step]); unif = list(uniforms[step])
        cache = {}

        def force(group=None):
            key = ('f', group)
            if key not in cache:
                cache[key] = chain.energy_forces(x, None if group is None else {int(group)})[1]
            return cache[key]

        def env(per_dof):
            e = dict(funcs); e.update(g)
            if per_dof:
                e.update(per); e.update(x=x, v=v, m=m)
            return e

        def evaluate(text, per_dof):
            e = env(per_dof)
            names = set(re.findall(r'[A-Za-z_][A-Za-z_0-9]*', text))
            if 'energy' in names:
                e['energy'] = chain.energy_forces(x)[0]
            for nme in names:
                if nme == 'f':
                    e['f'] = force()
                elif re.fullmatch(r'f\d+', nme):
                    e[nme] = force(int(nme[1:]))
            if 'gaussian' in names:
                e['gaussian'] = gauss.pop(0)
            if 'uniform' in names:
                e['uniform'] = unif.pop(0)
            return eval(_expr(text), {'__builtins__': {}}, e)

        pc, skip = 0, 0
        prog = rec.program
        while pc < len(prog):
            op = prog[pc]
            if skip:
                if op[0] == 'if': skip += 1
                elif op[0] == 'end': skip -= 1
                pc += 1
                continue
            if op[0] == 'per_dof':
                val = np.broadcast_to(np.asarray(evaluate(op[2], True), dtype=np.float64), (N, 3)).copy()
                if op[1] == 'x':
                    x = val; cache.clear()
                elif op[1] == 'v':
                    v = val
                else:
                    per[op[1]] = val
            elif op[0] == 'global':
                g[op[1]] = float(evaluate(op[2], False))
            elif op[0] == 'sum':
                g[op[1]] = float(np.sum(np.broadcast_to(evaluate(op[2], True), (N, 3))))
            elif op[0] == 'if':
                if not bool(eval(_EQ.sub('==', _expr(op[1])), {'__builtins__': {}}, env(False))):
                    skip = 1
            elif op[0] in ('end', 'update_context_state', 'constrain_positions', 'constrain_velocities'):
                pass                                   # (no constraints, no barostat / centre-of-mass remover in this system)
            else:
                raise ValueError(op)
            pc += 1
        assert not gauss and not unif, 'noise prepared for this step was not consumed'
        out.append(dict(x=x.tolist(), v=v.tolist(), **{k: g[k] for k in keep if k in g}))
    return out


def main():
    Langevin, lines = reference_class()
    chain = Chain()
    rng = np.random.default_rng(20260927)
    x0 = np.array([[0.0, 0.0, 0.0], [0.105, 0.02, -0.01], [0.16, 0.14, 0.03], [0.29, 0.15, 0.10]]) + rng.normal(0, 0.003, (4, 3))
    v0 = rng.normal(0, 0.4, (4, 3))
    _check_forces(chain, x0)
    out = dict(source='openmmtools/integrators.py:%d-%d (class LangevinIntegrator, executed)' % lines, kB=KB, seed=SEED, replica=REPLICA,
               temperature=TEMPERATURE, collision_rate=GAMMA, timestep=DT, n_steps=N_STEPS, x0=x0.tolist(), v0=v0.tolist(),
               masses=list(MASSES), bonds=[list(b) for b in BONDS], angles=[list(a) for a in ANGLES], cases=[])
    for splitting, shadow, heat, *rest in CASES:
        dt = rest[0] if rest else DT
        rec = Langevin(temperature=TEMPERATURE, collision_rate=GAMMA, timestep=dt, splitting=splitting, measure_shadow_work=shadow, measure_heat=heat)
        tokens = splitting.split()
        nO = tokens.count('O')
        noise, uniforms = [], []
        for step in range(N_STEPS):
            noise.append([md_oracle.gaussians3(SEED, md_oracle.STREAM_OU, np.arange(4), REPLICA, step * max(1, nO) + o) for o in range(nO)])
            us = []
            for brace in range(tokens.count('}')):
                w = md_oracle.draw(SEED, md_oracle.STREAM_METROPOLIS, brace, REPLICA, step)
                us.append(((int(w[2]) << 21) | (int(w[3]) >> 11)) / 9007199254740992.0)
            uniforms.append(us)
        traj = interpret(rec, chain, x0, v0, N_STEPS, noise, uniforms)
        out['cases'].append(dict(splitting=splitting, timestep=dt, measure_shadow_work=bool(rec._measure_shadow_work), measure_heat=bool(heat),
                                 program=rec.program, globals={k: rec.globals[k] for k in ('a', 'b', 'kT')},
                                 noise=[[n.tolist() for n in s] for s in noise], uniforms=uniforms, trajectory=traj))
        print('%-22s %2d program lines; a = %.12f b = %.12f; |x| after %d steps %.9f' % (
            splitting, len(rec.program), rec.globals['a'], rec.globals['b'], N_STEPS, float(np.abs(np.array(traj[-1]['x'])).sum())))
    # velocity Verlet and hybrid Monte Carlo: programs of their own in the reference (integrators.py:456-498, 885-1010); this package runs
    # them as the splittings 'V R V' and 'O { (V R V)^n }' of the same chain kernel -- the fixture holds what the reference's programs do
    others = other_reference_classes(['VelocityVerletIntegrator', 'HMCIntegrator'])
    out['other_integrators'] = []
    for name, kwargs in (('VelocityVerletIntegrator', dict(timestep=DT)), ('HMCIntegrator', dict(temperature=TEMPERATURE, nsteps=3, timestep=0.002)),
                         ('HMCIntegrator', dict(temperature=TEMPERATURE, nsteps=2, timestep=0.003))):
        cls, lines_ = others[name]
        rec = cls(**kwargs)
        if 'kT' not in rec.globals:
            rec.globals['kT'] = KB * TEMPERATURE                        # (velocity Verlet has no temperature; the interpreter wants the name)
        hmc = name == 'HMCIntegrator'
        noise = [[md_oracle.gaussians3(SEED, md_oracle.STREAM_OU, np.arange(4), REPLICA, step)] if hmc else [] for step in range(N_STEPS)]
        uniforms = []
        for step in range(N_STEPS):
            w = md_oracle.draw(SEED, md_oracle.STREAM_METROPOLIS, 0, REPLICA, step)
            uniforms.append([((int(w[2]) << 21) | (int(w[3]) >> 11)) / 9007199254740992.0] if hmc else [])
        traj = interpret(rec, chain, x0, v0, N_STEPS, noise, uniforms, keep=('naccept', 'ntrials', 'accept', 'Eold', 'Enew'))
        out['other_integrators'].append(dict(name=name, kwargs=kwargs, source='openmmtools/integrators.py:%d-%d' % lines_, program=rec.program,
                                             noise=[[n.tolist() for n in s_] for s_ in noise], uniforms=uniforms, trajectory=traj))
        print('%-26s %-44s %2d program lines; accepted %s' % (name, kwargs, len(rec.program), [int(t['accept']) for t in traj] if hmc else '-'))
    with open(OUT, 'w') as fh:
        json.dump(out, fh, separators=(',', ':'))
    print('wrote', OUT, os.path.getsize(OUT), 'bytes')


if __name__ == '__main__':
    main()

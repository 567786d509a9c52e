"""Golden values of the alchemical GBSA (OBC2) energy, from the reference's own CustomGBForce expression strings.

/root/reference/openmmtools/alchemy/alchemy.py:2144-2225 (_alchemically_modify_GBSAOBCForce) hands OpenMM a CustomGBForce made of string
literals: the computed values I (a sum over particle pairs) and B (per particle) and three energy terms.  openmm is absent here; this
script takes those literals out of the function's syntax tree UNCHANGED and evaluates them with a small interpreter of the CustomGBForce
semantics (OpenMM user guide, "CustomGBForce"):
    computed value of type ParticlePairNoExclusions:  value_i = sum over j != i of expr(r_ij; parameters of i as ..1, of j as ..2)
    computed value of type SingleParticle:            value_i = expr(parameters and earlier computed values of i)
    energy term SingleParticle:                       sum_i expr(i);     ParticlePairNoExclusions: sum over pairs i < j of expr(r; ..1, ..2)
on random small systems at several lambda_electrostatics; expressions, inputs and energies go to tests/golden/reference_gbsa.json.
/root/reference does not exist on the GPU box: the tests read only the JSON.     usage: python tests/golden/make_golden_gbsa.py
"""
import ast
import json
import math
import os
import re

import numpy as np

REF = '/root/reference/openmmtools/alchemy/alchemy.py'
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'reference_gbsa.json')


def reference_strings():
    tree = ast.parse(open(REF).read())
    cls = next(n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == 'AbsoluteAlchemicalFactory')
    fn = next(n for n in cls.body if isinstance(n, ast.FunctionDef) and n.name == '_alchemically_modify_GBSAOBCForce')
    computed, energy, globals_ = [], [], {}
    for node in ast.walk(fn):
        if not (isinstance(node, ast.Call) and isinstance(node.func, ast.Attribute)):
            continue
        kind = lambda a: a.attr if isinstance(a, ast.Attribute) else None
        if node.func.attr == 'addComputedValue':
            computed.append((node.lineno, node.args[0].value, node.args[1].value, kind(node.args[2])))
        elif node.func.attr == 'addEnergyTerm':
            energy.append((node.lineno, node.args[0].value, kind(node.args[1])))
        elif node.func.attr == 'addGlobalParameter' and isinstance(node.args[1], ast.Constant):
            globals_[node.args[0].value] = float(node.args[1].value)
    return sorted(computed), sorted(energy), globals_, (fn.lineno, fn.end_lineno)


class _Names(dict):
    def __init__(self, values, definitions):
        super().__init__(values)
        self._defs = definitions
    def __missing__(self, name):
        if name not in self._defs:
            raise KeyError(name)
        self[name] = eval(self._defs[name], {'__builtins__': {}}, self)
        return self[name]


FUNCS = dict(sqrt=math.sqrt, exp=math.exp, log=math.log, tanh=math.tanh, abs=abs, max=max, min=min, step=lambda x: 1.0 if x >= 0 else 0.0)


def evaluate(expression, values):
    expression = re.sub(r'\bor\b', 'or_', expression)               # (OpenMM's variable `or` is a keyword of the language this interpreter borrows)
    parts = [p.strip() for p in expression.split(';') if p.strip()]
    defs = {}
    for p in parts[1:]:
        name, rhs = p.split('=', 1)
        defs[name.strip()] = rhs.replace('^', '**')
    return float(eval(parts[0].replace('^', '**'), {'__builtins__': {}}, _Names(dict(values, **FUNCS), defs)))


def custom_gb_energy(computed, energy, globals_, x, per_particle):
    """per_particle: dict name -> array.  Returns the energy (kJ/mol)."""
    n = len(x)
    values = {}
    def of(i, suffix=''):
        d = {k + suffix: float(v[i]) for k, v in per_particle.items()}
        d.update({k + suffix: float(v[i]) for k, v in values.items()})
        return d
    for _, name, expr, kind in computed:
        out = np.zeros(n)
        for i in range(n):
            if kind == 'SingleParticle':
                out[i] = evaluate(expr, dict(globals_, **of(i)))
            else:
                for j in range(n):
                    if j != i:
                        out[i] += evaluate(expr, dict(globals_, r=float(np.linalg.norm(x[i] - x[j])), **of(i, '1'), **of(j, '2')))
        values[name] = out
    e = 0.0
    for _, expr, kind in energy:
        if kind == 'SingleParticle':
            e += sum(evaluate(expr, dict(globals_, **of(i))) for i in range(n))
        else:
            e += sum(evaluate(expr, dict(globals_, r=float(np.linalg.norm(x[i] - x[j])), **of(i, '1'), **of(j, '2'))) for i in range(n) for j in range(i + 1, n))
    return e, {k: v.tolist() for k, v in values.items()}


def main():
    computed, energy, globals_, lines = reference_strings()
    out = {'source': 'alchemy.py:%d-%d' % lines, 'computed_values': [c[1:] for c in computed], 'energy_terms': [e[1:] for e in energy],
           'globals_in_the_function': globals_, 'cases': []}
    rng = np.random.default_rng(20261001)
    for n in (5, 9):
        x = rng.uniform(0.0, 0.9, size=(n, 3))
        # keep atoms apart (real radii: 0.12 - 0.2 nm)
        while True:
            d = np.linalg.norm(x[:, None] - x[None], axis=-1) + np.eye(n)
            if d.min() > 0.11:
                break
            x = rng.uniform(0.0, 0.9, size=(n, 3))
        pp = dict(charge=rng.uniform(-0.8, 0.8, n), radius=rng.uniform(0.12, 0.2, n), scale=rng.uniform(0.7, 0.9, n),
                  alchemical=(np.arange(n) < 2).astype(float))
        for lam in (1.0, 0.6, 0.0):
            g = dict(globals_, lambda_electrostatics=lam, solventDielectric=78.5, soluteDielectric=1.0)
            e, vals = custom_gb_energy(computed, energy, g, x, pp)
            out['cases'].append(dict(x=x.tolist(), charge=pp['charge'].tolist(), radius=pp['radius'].tolist(), scale=pp['scale'].tolist(),
                                     alchemical=pp['alchemical'].tolist(), lambda_electrostatics=lam, solventDielectric=78.5, soluteDielectric=1.0,
                                     energy=e, I=vals['I'], B=vals['B']))
    with open(OUT, 'w') as fh:
        json.dump(out, fh, separators=(',', ':'))
    for c in out['computed_values'] + out['energy_terms']:
        print(c)
    print([c['energy'] for c in out['cases']])


if __name__ == '__main__':
    main()

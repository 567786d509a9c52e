"""Golden vectors for the soft-core energy EXPRESSIONS, produced from the reference's own string literals.

The alchemical factory of the reference builds its CustomNonbondedForce / CustomBondForce energies from string constants inside

    /root/reference/openmmtools/alchemy/alchemy.py:1356-1390   _get_sterics_energy_expressions
    /root/reference/openmmtools/alchemy/alchemy.py:1392-1471   _get_electrostatics_energy_expressions
    /root/reference/openmmtools/alchemy/alchemy.py:1473-1508   _get_reaction_field_unique_expression
    /root/reference/openmmtools/alchemy/alchemy.py:1510-1537   _get_pme_direct_space_unique_expression

openmm is absent here, so the package cannot be imported; but these four methods only format strings.  This script takes their
definitions out of the module's syntax tree UNCHANGED, executes them on a stand-in ``self`` (the factory's option attributes) and
a stand-in NonbondedForce (method, cutoff, dielectric, Ewald tolerance), and so obtains the literal expressions the reference
would hand to OpenMM.  Each expression is then evaluated by a small interpreter of OpenMM's expression syntax
('value; name = definition; ...', ^ = power) -- the same one tests/test_alchemical_store_cpu.py uses on the documents this package
writes -- on a grid of (r, sigma, epsilon, charges, lambda); expression strings, inputs and values go to
tests/golden/reference_alchemy_expressions.json.

Injected (the only things that are not the reference's text): ``openmm.NonbondedForce`` method constants (OpenMM's enum order),
a four-operation Quantity for the reaction-field constants, and ONE_4PI_EPS0 -- the reference computes it with openmm.unit from the
literals of openmmtools/constants.py:12-14 (E_CHARGE 1.602176634e-19 C, EPSILON0 8.8541878128e-12 F/m, Avogadro's number); the
same arithmetic is done here in plain floats.

/root/reference does not exist on the GPU box: the tests read only the JSON.   usage: python tests/golden/make_golden_alchemy_strings.py
"""
import ast
import json
import os
from math import pi

import numpy as np
from scipy.special import erfc

REF = '/root/reference/openmmtools/alchemy/alchemy.py'
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'reference_alchemy_expressions.json')
WANTED = ('_get_sterics_energy_expressions', '_get_electrostatics_energy_expressions', '_get_reaction_field_unique_expression',
          '_get_pme_direct_space_unique_expression')

# openmmtools/constants.py:12-14 in plain floats (AVOGADRO_CONSTANT_NA of openmm.unit = 6.02214076e23)
E_CHARGE, AVOGADRO = 1.602176634e-19, 6.02214076e23
EPSILON0 = 1e-6 * 8.8541878128e-12 / (AVOGADRO * E_CHARGE ** 2)
ONE_4PI_EPS0 = 1 / (4 * pi * EPSILON0)


# ---- stand-ins ----------------------------------------------------------------------------------------------------------
class Q:
    """value * nm^e: what the reaction-field / Ewald constants need of openmm.unit.Quantity"""
    def __init__(self, v, e=0):
        self.v, self.e = float(v), e
    unit = property(lambda self: Q(1.0, self.e))
    def __pow__(self, p): return Q(self.v ** p, self.e * p)
    def __mul__(self, o): return Q(self.v * o.v, self.e + o.e) if isinstance(o, Q) else Q(self.v * o, self.e)
    __rmul__ = __mul__
    def __truediv__(self, o): return Q(self.v / o.v, self.e - o.e) if isinstance(o, Q) else Q(self.v / o, self.e)
    def __rtruediv__(self, o): return Q(o / self.v, -self.e)
    def __eq__(self, o): return (self.v == o) if not isinstance(o, Q) else (self.v, self.e) == (o.v, o.e)
    def value_in_unit_system(self, system): return self.v


class _Unit:
    nanometers = Q(1.0, 1)
    md_unit_system = 'md'


class _NonbondedForceEnum:
    NoCutoff, CutoffNonPeriodic, CutoffPeriodic, Ewald, PME = range(5)      # OpenMM's NonbondedForce::NonbondedMethod


class _OpenMM:
    NonbondedForce = _NonbondedForceEnum


class ReferenceForce:
    def __init__(self, method, cutoff=1.0, dielectric=78.3, tolerance=5e-4, alpha=0.0):
        self.method, self.cutoff, self.dielectric, self.tolerance, self.alpha = method, cutoff, dielectric, tolerance, alpha
    def getNonbondedMethod(self): return self.method
    def getReactionFieldDielectric(self): return self.dielectric
    def getCutoffDistance(self): return Q(self.cutoff, 1)
    def getPMEParameters(self): return [Q(self.alpha, -1), 0, 0, 0]
    def getEwaldErrorTolerance(self): return self.tolerance


class Factory:
    """the option attributes the four methods read (alchemy.py:626-635 has the defaults)"""
    def __init__(self, pme='exact', rf='switched', consistent=False):
        self.alchemical_pme_treatment, self.alchemical_rf_treatment, self.consistent_exceptions = pme, rf, consistent


def reference_methods():
    tree = ast.parse(open(REF).read())
    cls = next(n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == 'AbsoluteAlchemicalFactory')
    fns = [n for n in cls.body if isinstance(n, ast.FunctionDef) and n.name in WANTED]
    assert len(fns) == len(WANTED)
    for f in fns:
        f.decorator_list = []
    ns = dict(openmm=_OpenMM, unit=_Unit, np=np, ONE_4PI_EPS0=ONE_4PI_EPS0)
    exec(compile(ast.Module(body=fns, type_ignores=[]), REF, 'exec'), ns)
    for name in WANTED:
        setattr(Factory, name, ns[name])
    return {f.name: (f.lineno, f.end_lineno) for f in fns}


# ---- OpenMM's expression syntax -------------------------------------------------------------------------------------------
class _Names(dict):
    def __init__(self, values, definitions):
        super().__init__(values)
        self._defs = definitions
    def __missing__(self, name):
        if name not in self._defs:
            raise KeyError(name)
        self[name] = eval(self._defs[name], {'__builtins__': {}}, self)
        return self[name]


def evaluate(expression, values):
    parts = [p.strip() for p in expression.split(';') if p.strip()]
    defs = {}
    for p in parts[1:]:
        name, rhs = p.split('=', 1)
        defs[name.strip()] = rhs.replace('^', '**')
    return float(eval(parts[0].replace('^', '**'), {'__builtins__': {}}, _Names(dict(values, sqrt=np.sqrt, erfc=erfc), defs)))


SOFTCORE = dict(softcore_alpha=0.5, softcore_beta=0.0, softcore_a=1.0, softcore_b=1.0, softcore_c=6.0, softcore_d=1.0, softcore_e=1.0,
                softcore_f=2.0)                                   # AlchemicalRegion defaults, alchemy.py:417-427


def main():
    lines = reference_methods()
    out = {'source': {k: 'alchemy.py:%d-%d' % v for k, v in lines.items()}, 'ONE_4PI_EPS0': ONE_4PI_EPS0, 'softcore': SOFTCORE,
           'expressions': {}, 'samples': {}}
    E = out['expressions']
    f = Factory()
    mix, exc = f._get_sterics_energy_expressions([''])
    E['sterics_mixing_rules'], E['sterics_exception'] = mix, exc
    E['sterics_pair'] = exc + mix                                # alchemy.py:1710: the CustomNonbondedForce energy
    E['sterics_exception_two_regions'] = f._get_sterics_energy_expressions(['_zero', '_one'])[1]
    NB = _NonbondedForceEnum
    E['electrostatics_nocutoff'], E['electrostatics_exception_nocutoff'] = f._get_electrostatics_energy_expressions(ReferenceForce(NB.NoCutoff), [''])
    E['electrostatics_rf_switched'], E['electrostatics_exception_rf'] = f._get_electrostatics_energy_expressions(ReferenceForce(NB.CutoffPeriodic), [''])
    E['electrostatics_rf_shifted'] = Factory(rf='shifted')._get_electrostatics_energy_expressions(ReferenceForce(NB.CutoffPeriodic), [''])[0]
    E['electrostatics_pme_direct_space'] = Factory(pme='direct-space')._get_electrostatics_energy_expressions(ReferenceForce(NB.PME), [''])[0]
    E['electrostatics_pme_coulomb'] = Factory(pme='coulomb')._get_electrostatics_energy_expressions(ReferenceForce(NB.PME), [''])[0]
    E['electrostatics_exception_rf_consistent'] = Factory(consistent=True)._get_electrostatics_energy_expressions(ReferenceForce(NB.CutoffPeriodic), [''])[1]

    # sterics: a grid the engines can reproduce pair by pair (5 parameter pairs x 5 distances below the switching distance x 4 lambdas)
    rng = np.random.default_rng(20260926)
    params = [(float(s), float(e)) for s, e in zip(rng.uniform(0.25, 0.40, 5), rng.uniform(0.2, 1.2, 5))]
    grid = []
    for sigma, epsilon in params:
        for r in rng.uniform(0.12, 0.80, 5):
            for lam in (1.0, 0.7, 0.35, 0.0):
                v = dict(r=float(r), sigma1=sigma, sigma2=sigma, epsilon1=epsilon, epsilon2=epsilon, lambda_sterics=lam)
                grid.append(dict(v, value=evaluate(E['sterics_pair'], dict(v, **SOFTCORE)),
                                 exception_value=evaluate(E['sterics_exception'], dict(SOFTCORE, r=float(r), sigma=sigma, epsilon=epsilon, lambda_sterics=lam))))
    out['samples']['sterics'] = grid
    # non-default exponents too (what a user may set on the region): 100 random points, interpreter values only
    rnd = []
    for _ in range(100):
        sc = dict(SOFTCORE, softcore_alpha=float(rng.uniform(0.2, 0.8)), softcore_a=float(rng.choice([1.0, 2.0])),
                  softcore_b=float(rng.choice([1.0, 2.0])), softcore_c=float(rng.choice([6.0, 12.0])))
        v = dict(r=float(rng.uniform(0.05, 1.2)), sigma1=float(rng.uniform(0.1, 0.45)), sigma2=float(rng.uniform(0.1, 0.45)),
                 epsilon1=float(rng.uniform(0.05, 2.0)), epsilon2=float(rng.uniform(0.05, 2.0)), lambda_sterics=float(rng.uniform(0, 1)))
        rnd.append(dict(v, softcore=sc, value=evaluate(E['sterics_pair'], dict(v, **sc))))
    out['samples']['sterics_random'] = rnd
    for key in ('electrostatics_nocutoff', 'electrostatics_rf_switched', 'electrostatics_rf_shifted', 'electrostatics_pme_direct_space',
                'electrostatics_pme_coulomb'):
        pts = []
        for _ in range(40):
            sc = dict(SOFTCORE, softcore_beta=float(rng.choice([0.0, 0.5])))
            v = dict(r=float(rng.uniform(0.05, 1.0)), sigma1=float(rng.uniform(0.1, 0.45)), sigma2=float(rng.uniform(0.1, 0.45)),
                     charge1=float(rng.uniform(-1, 1)), charge2=float(rng.uniform(-1, 1)), lambda_electrostatics=float(rng.uniform(0, 1)))
            pts.append(dict(v, softcore_beta=sc['softcore_beta'], value=evaluate(E[key], dict(v, **sc))))
        out['samples'][key] = pts
    with open(OUT, 'w') as fh:
        json.dump(out, fh, separators=(',', ':'))
    print('wrote', OUT, {k: len(v) for k, v in out['samples'].items()})
    for k, v in E.items():
        print('%-40s %s' % (k, v[:110]))


if __name__ == '__main__':
    main()

"""Constructor defaults of the benchmark test systems, taken from the reference's own source (parity of the WORKLOADS: BASELINE.json's
configs are 'testsystems.X()' with default arguments).

The reference's openmmtools/testsystems.py imports OpenMM, absent here; the defaults are literals in the `__init__` signatures.  This
script takes the signatures of the classes the five configs use out of the module's syntax tree, evaluates every default expression in a
namespace that holds a unit table written out here (plain floats in the MD unit system: nm, ps, amu, kJ/mol, K -- NOT this package's
`unit` module, which the test thereby checks as well), the three module-level DEFAULT_* constants evaluated the same way, and name
stand-ins for `app.HBonds` / `app.PME`, and writes source text and value per
argument to tests/golden/reference_testsystem_defaults.json.  tests/test_testsystem_defaults.py compares this package's signatures.

usage: python tests/golden/make_golden_testsystem_defaults.py        (/root/reference is not needed by the tests)"""
import ast
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))


class unit:
    """the units the signatures use, as factors to the MD unit system (openmm.unit.md_unit_system)"""
    nanometers = nanometer = 1.0
    angstroms = angstrom = 0.1
    amu = daltons = dalton = 1.0
    kilojoules_per_mole = kilojoule_per_mole = 1.0
    kilocalories_per_mole = kilocalorie_per_mole = 4.184                 # thermochemical calorie
    kelvin = kelvins = 1.0
    picoseconds = picosecond = 1.0
    femtoseconds = femtosecond = 1.0e-3
    atmospheres = atmosphere = 101325.0 * 6.02214076e23 * 1.0e-30         # Pa = J / m^3 -> kJ / mol / nm^3
    elementary_charge = 1.0

REF = '/root/reference/openmmtools/testsystems.py'
OUT = os.path.join(HERE, 'reference_testsystem_defaults.json')
CLASSES = ('HarmonicOscillator', 'LennardJonesFluid', 'IdealGas', 'AlanineDipeptideExplicit', 'HostGuestExplicit', 'DHFRExplicit')


class _App:
    HBonds, PME, CutoffPeriodic, NoCutoff = 'HBonds', 'PME', 'CutoffPeriodic', 'NoCutoff'


def main():
    tree = ast.parse(open(REF).read())
    ns = dict(unit=unit, app=_App, None_=None)
    constants = {}
    for node in tree.body:
        if isinstance(node, ast.Assign) and len(node.targets) == 1 and isinstance(node.targets[0], ast.Name) and node.targets[0].id.startswith('DEFAULT_'):
            name = node.targets[0].id
            src = ast.unparse(node.value)
            ns[name] = eval(compile(ast.Expression(node.value), REF, 'eval'), dict(ns))
            constants[name] = dict(source=src, value=ns[name], line=node.lineno)
    out = dict(source=REF.replace('/root/reference/', ''), units='MD unit system: nm, ps, amu, kJ/mol, K, elementary charge', constants=constants, classes={})
    for node in tree.body:
        if isinstance(node, ast.ClassDef) and node.name in CLASSES:
            init = next(n for n in node.body if isinstance(n, ast.FunctionDef) and n.name == '__init__')
            a = init.args
            names = [x.arg for x in a.args][1:]                           # without self
            defaults = [None] * (len(names) - len(a.defaults)) + list(a.defaults)
            args = {}
            for name, d in zip(names, defaults):
                if d is None:
                    args[name] = dict(source=None, required=True)
                    continue
                src = ast.unparse(d)
                try:
                    val = eval(compile(ast.Expression(d), REF, 'eval'), dict(ns))
                    if not isinstance(val, (int, float, str, bool, type(None))):
                        val = repr(val)
                    args[name] = dict(source=src, value=val)
                except Exception as exc:                                   # an expression this namespace cannot evaluate: keep the text
                    args[name] = dict(source=src, unevaluated=str(exc))
            files = sorted({c.args[0].value for c in ast.walk(init) if isinstance(c, ast.Call) and getattr(c.func, 'id', '') == 'get_data_filename'
                            and c.args and isinstance(c.args[0], ast.Constant)})
            # HostGuestExplicit passes createSystem's arguments through a `defaults` dictionary inside __init__ (testsystems.py:3828-3833)
            inner = {}
            for stmt in ast.walk(init):
                if isinstance(stmt, ast.Assign) and isinstance(stmt.targets[0], ast.Name) and stmt.targets[0].id == 'defaults' and isinstance(stmt.value, ast.Dict):
                    for k, v in zip(stmt.value.keys, stmt.value.values):
                        try:
                            inner[k.value] = dict(source=ast.unparse(v), value=eval(compile(ast.Expression(v), REF, 'eval'), dict(ns)))
                        except Exception:
                            inner[k.value] = dict(source=ast.unparse(v))          # (refers to an argument of __init__)
            out['classes'][node.name] = dict(line=init.lineno, arguments=args, data_files=files, create_system_defaults=inner)
    with open(OUT, 'w') as fh:
        json.dump(out, fh, indent=1, sort_keys=True)
    for k, v in out['classes'].items():
        print(k, {n: d.get('value', d.get('source')) for n, d in v['arguments'].items()}, v['data_files'])


if __name__ == '__main__':
    main()

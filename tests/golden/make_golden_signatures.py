"""Public signatures of the classes on the hot path's boundary, taken from the reference's source (the mirror keeps the
reference's names, argument meaning and defaults so that a user's script runs unchanged).

openmm is absent here, so the reference modules cannot be imported; signatures are syntax.  For every class of the reference modules
listed in MODULES, this script takes the argument names and default expressions of `__init__` and of every public method (properties
aside) out of the syntax tree, evaluates each default in a namespace with a unit table written out here
(MD unit system: nm, ps, amu, kJ/mol, K), numpy and the module's own simple constants, and stores source text + value (when the
expression evaluates to a plain number / string / bool / None) in tests/golden/reference_signatures.json.
tests/test_signatures.py compares this package's signatures.     usage: python tests/golden/make_golden_signatures.py"""
import ast
import json
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, 'reference_signatures.json')
ROOT = '/root/reference/openmmtools/'
MODULES = {'integrators': 'integrators.py', 'mcmc': 'mcmc.py', 'states': 'states.py', 'alchemy': 'alchemy/alchemy.py',
           'multistate.multistatesampler': 'multistate/multistatesampler.py', 'multistate.replicaexchange': 'multistate/replicaexchange.py',
           'multistate.sams': 'multistate/sams.py', 'multistate.paralleltempering': 'multistate/paralleltempering.py',
           'multistate.multistatereporter': 'multistate/multistatereporter.py'}
METHODS = ('__init__', 'create', 'run', 'extend', 'minimize', 'equilibrate', 'apply', 'from_storage', 'create_alchemical_system')


class unit:
    nanometers = nanometer = 1.0
    angstroms = angstrom = 0.1
    amu = daltons = dalton = 1.0
    kilojoules_per_mole = kilojoule_per_mole = 1.0
    kilocalories_per_mole = kilocalorie_per_mole = 4.184
    kelvin = kelvins = 1.0
    picoseconds = picosecond = 1.0
    femtoseconds = femtosecond = 1.0e-3
    atmospheres = atmosphere = 101325.0 * 6.02214076e23 * 1.0e-30
    elementary_charge = 1.0
    dimensionless = 1.0


def _is_property(fn):
    return any((isinstance(d, ast.Name) and d.id == 'property') or (isinstance(d, ast.Attribute) and d.attr in ('setter', 'getter', 'deleter')) for d in fn.decorator_list)


def signature(fn, ns):
    a = fn.args
    names = [x.arg for x in a.args]
    if names and names[0] in ('self', 'cls'):
        names = names[1:]
    defaults = [None] * (len(names) - len(a.defaults)) + list(a.defaults)
    kwonly = [(x.arg, d) for x, d in zip(a.kwonlyargs, a.kw_defaults)]
    out = []
    for name, d in list(zip(names, defaults)) + kwonly:
        rec = dict(name=name)
        if d is None:
            rec['required'] = True
        else:
            rec['source'] = ast.unparse(d)
            try:
                val = eval(compile(ast.Expression(d), 'reference', 'eval'), dict(ns))
                if isinstance(val, (bool, int, float, str, type(None))):
                    rec['value'] = val
                elif isinstance(val, (np.floating, np.integer)):
                    rec['value'] = float(val)
            except Exception:
                pass                                     # (an expression over names only the imported module has: the text is kept)
        out.append(rec)
    return dict(arguments=out, var_positional=a.vararg.arg if a.vararg else None, var_keyword=a.kwarg.arg if a.kwarg else None)


def main():
    out = dict(units='MD unit system: nm, ps, amu, kJ/mol, K', modules={})
    n_cls = n_sig = 0
    for mod, rel in MODULES.items():
        tree = ast.parse(open(ROOT + rel).read())
        ns = dict(unit=unit, np=np, numpy=np)
        for node in tree.body:                         # simple module-level constants (e.g. _DEFAULT_..., kB stays out: needs openmm)
            if isinstance(node, ast.Assign) and len(node.targets) == 1 and isinstance(node.targets[0], ast.Name):
                try:
                    ns[node.targets[0].id] = eval(compile(ast.Expression(node.value), 'reference', 'eval'), dict(ns))
                except Exception:
                    pass
        classes = {}
        for node in tree.body:
            if not isinstance(node, ast.ClassDef) or node.name.startswith('_'):
                continue
            sigs = {}
            for fn in node.body:
                if isinstance(fn, ast.FunctionDef) and (fn.name in METHODS or not fn.name.startswith('_')) and not _is_property(fn):
                    sigs[fn.name] = dict(signature(fn, ns), line=fn.lineno)
                    n_sig += 1
            # the coded errors of states.py: the tuple of names unpacked from range(n) and the message table
            names, messages = None, None
            for stmt in node.body:
                if isinstance(stmt, ast.Assign) and isinstance(stmt.targets[0], ast.Tuple) and isinstance(stmt.value, ast.Call) and getattr(stmt.value.func, 'id', '') == 'range':
                    names = [e.id for e in stmt.targets[0].elts]
                if isinstance(stmt, ast.Assign) and getattr(stmt.targets[0], 'id', '') == 'error_messages' and isinstance(stmt.value, ast.Dict):
                    messages = {k.id: v.value for k, v in zip(stmt.value.keys, stmt.value.values)}
            # public properties and class-level descriptors (the reference's _StoredProperty options): names only
            props = sorted({n.name for n in node.body if isinstance(n, ast.FunctionDef) and not n.name.startswith('_')
                            and any(isinstance(d, ast.Name) and d.id == 'property' for d in n.decorator_list)} |
                           {t.id for n in node.body if isinstance(n, ast.Assign) and isinstance(n.value, ast.Call) for t in n.targets
                            if isinstance(t, ast.Name) and not t.id.startswith('_')})
            if props:
                sigs['properties'] = props
            if names and messages:
                sigs['error_codes'] = [dict(name=n, number=i, message=messages[n]) for i, n in enumerate(names)]
            if sigs:
                classes[node.name] = sigs
                n_cls += 1
        out['modules'][mod] = dict(file='openmmtools/' + rel, classes=classes)
    with open(OUT, 'w') as fh:
        json.dump(out, fh, separators=(',', ':'), sort_keys=True)
    print('wrote', OUT, n_cls, 'classes', n_sig, 'signatures')


if __name__ == '__main__':
    main()

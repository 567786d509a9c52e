"""Golden literals of the softened bonded custom forces and of the unshifted reaction field, taken out of the reference's syntax tree.

    /root/reference/openmmtools/alchemy/alchemy.py  _alchemically_modify_PeriodicTorsionForce / HarmonicAngleForce / HarmonicBondForce
        (energy_function f-strings, the per-term parameter names, the lambda's base name)
    /root/reference/openmmtools/forces.py           UnshiftedReactionFieldForce.__init__ (the constant pieces of its energy expression)

openmm is absent here, so nothing is executed but the f-strings themselves (their only free name is ``lambda_variable_name``).
/root/reference does not exist on the GPU box: the tests read only tests/golden/reference_bonded_expressions.json.
usage: python tests/golden/make_golden_bonded_strings.py
"""
import ast
import json
import os

ALCHEMY = '/root/reference/openmmtools/alchemy/alchemy.py'
FORCES = '/root/reference/openmmtools/forces.py'
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'reference_bonded_expressions.json')


def _function(tree, cls_name, fn_name):
    cls = next(n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == cls_name)
    return next(n for n in cls.body if isinstance(n, ast.FunctionDef) and n.name == fn_name)


def main():
    out = {'source': {}, 'bonded': {}, 'unshifted_reaction_field': {}}
    tree = ast.parse(open(ALCHEMY).read())
    for kind, fn_name in (('torsion', '_alchemically_modify_PeriodicTorsionForce'), ('angle', '_alchemically_modify_HarmonicAngleForce'),
                          ('bond', '_alchemically_modify_HarmonicBondForce')):
        fn = _function(tree, 'AbsoluteAlchemicalFactory', fn_name)
        out['source'][kind] = 'alchemy.py:%d-%d' % (fn.lineno, fn.end_lineno)
        energy = base = None
        per = []
        for node in ast.walk(fn):
            if isinstance(node, ast.Assign) and isinstance(node.targets[0], ast.Name):
                if node.targets[0].id == 'energy_function':
                    energy = eval(compile(ast.Expression(node.value), ALCHEMY, 'eval'), {'lambda_variable_name': 'LAMBDA'})
                if node.targets[0].id == 'lambda_variable_name' and isinstance(node.value, ast.Constant):
                    base = node.value.value
            if isinstance(node, ast.Call) and isinstance(node.func, ast.Attribute) and node.func.attr.startswith('addPer') and node.args and isinstance(node.args[0], ast.Constant):
                per.append((node.lineno, node.args[0].value))
        out['bonded'][kind] = dict(energy=energy, lambda_base_name=base, per_term_parameters=[p for _, p in sorted(per)])
    tree = ast.parse(open(FORCES).read())
    cls = next(n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == 'UnshiftedReactionFieldForce')
    init = next(n for n in cls.body if isinstance(n, ast.FunctionDef) and n.name == '__init__')
    out['source']['unshifted_reaction_field'] = 'forces.py:%d-%d' % (init.lineno, init.end_lineno)
    pieces = []
    for node in ast.walk(init):
        if isinstance(node, (ast.Assign, ast.AugAssign)) and getattr(node.targets[0] if isinstance(node, ast.Assign) else node.target, 'id', None) == 'energy_expression':
            if isinstance(node.value, ast.Constant):
                pieces.append((node.lineno, node.value.value))
            elif isinstance(node.value, ast.JoinedStr):               # f"k_rf = {...:f};": the constant head of the piece
                pieces.append((node.lineno, ''.join(v.value for v in node.value.values if isinstance(v, ast.Constant))))
    out['unshifted_reaction_field']['energy_pieces'] = [p for _, p in sorted(pieces)]
    per = [n.args[0].value for n in ast.walk(init) if isinstance(n, ast.Call) and isinstance(n.func, ast.Attribute) and n.func.attr == 'addPerParticleParameter']
    out['unshifted_reaction_field']['per_particle_parameters'] = per
    with open(OUT, 'w') as fh:
        json.dump(out, fh, indent=1)
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()

"""Golden vectors produced by the reference's own code for the rotation proposal of MCRotationMove (mcmc.py:1842-1906).

``_rotation_matrix_from_quaternion`` and ``_generate_uniform_quaternion`` are plain numpy; the reference package cannot be
imported here (openmm is absent), so the two function definitions are taken out of the module's syntax tree unchanged (only the
``@staticmethod`` decorators are dropped), compiled and run with numpy's global stream seeded.  Output:
tests/golden/mc_rotation_reference.json.     usage: python tests/golden/make_golden_mc_moves.py"""
import ast
import json
import os
import numpy as np

SRC = '/root/reference/openmmtools/mcmc.py'
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'mc_rotation_reference.json')


def reference_functions():
    tree = ast.parse(open(SRC).read())
    cls = next(n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == 'MCRotationMove')
    ns = {'np': np}
    for node in cls.body:
        if isinstance(node, ast.FunctionDef) and node.name in ('_rotation_matrix_from_quaternion', '_generate_uniform_quaternion'):
            node.decorator_list = []
            exec(compile(ast.Module(body=[node], type_ignores=[]), SRC, 'exec'), ns)
    return ns['_rotation_matrix_from_quaternion'], ns['_generate_uniform_quaternion']


if __name__ == '__main__':
    to_matrix, quaternion = reference_functions()
    cases = []
    for seed in range(8):
        np.random.seed(seed)
        q = quaternion()
        cases.append(dict(seed=seed, quaternion=[float(c) for c in q], matrix=np.asarray(to_matrix(q)).tolist()))
    for q in ([0.0, 0.0, 0.0, 0.0], [2.0, 0.0, 0.0, 0.0], [0.3, -1.2, 0.5, 2.0]):           # zero norm and unnormalised inputs
        cases.append(dict(seed=None, quaternion=q, matrix=np.asarray(to_matrix(np.array(q))).tolist()))
    with open(OUT, 'w') as fh:
        json.dump(dict(source='openmmtools/mcmc.py:1842-1906 executed from /root/reference', cases=cases), fh, indent=1)
    print(len(cases), 'cases ->', OUT)

"""What the reference's own parser says about Langevin splitting strings (integrators.py:1319-1402 _sanity_check /
_verify_metropolization, :1474-1537 _parse_splitting_string): the three methods are taken from the class's syntax tree unchanged
and run on a stand-in ``self`` that only owns the dispatch-table keys (in the reference's order, :1164-1170).  Output:
tests/golden/splittings_reference.json -- per string either the parse result or the type of the exception raised.
usage: python tests/golden/make_golden_splittings.py"""
import ast
import json
import os
import re

SRC = '/root/reference/openmmtools/integrators.py'
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'splittings_reference.json')

STRINGS = [
    'V R O R V', 'O V R V O', 'V R R O R R V', 'R V O V R', 'O { V R V } O', 'v r o r v',
    'V0 V1 R R O R R V1 R R O R R V1 V0', 'V0 R O R V0', 'V1 V0 R O R V0 V1', 'V0 V R O R V1', 'V R O R V1',
    'V31 R O R V31', 'V32 R O R V32', 'Vx R O R V', 'V R V', 'R O R', 'V O V', 'V R O R V X', 'V R Q O', 'V  R O R V',
    'O { V R V O', '{ V R V } O', 'V { R } V O', 'R { V } O', 'O { V { R } V } O', '{ V { R } V } O', 'O } V R V { O',
    'O { V R O R V }', 'OR V', 'V R O 12', 'O {' + ' V R V' * 3 + ' }',
]


class Stub:
    _step_dispatch_table = {'O': None, 'R': None, '{': None, '}': None, 'V': None}


def reference_methods():
    tree = ast.parse(open(SRC).read())
    cls = next(n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == 'LangevinIntegrator')
    ns = {'re': re}
    for node in cls.body:
        if isinstance(node, ast.FunctionDef) and node.name in ('_sanity_check', '_verify_metropolization', '_parse_splitting_string'):
            node.decorator_list = []
            exec(compile(ast.Module(body=[node], type_ignores=[]), SRC, 'exec'), ns)
            setattr(Stub, node.name, ns[node.name])


if __name__ == '__main__':
    reference_methods()
    cases = []
    for s in STRINGS:
        try:
            counts, mts, n_v = Stub()._parse_splitting_string(s)
            cases.append(dict(splitting=s, ok=True, counts=counts, mts=mts, n_v=n_v))
        except Exception as exc:                                   # noqa: BLE001 -- the TYPE is the datum
            cases.append(dict(splitting=s, ok=False, error=type(exc).__name__, message=str(exc)))
    with open(OUT, 'w') as fh:
        json.dump(dict(source='openmmtools/integrators.py:1319-1402, 1474-1537 executed from /root/reference', cases=cases), fh, indent=1)
    for c in cases:
        print('%-40r %s' % (c['splitting'][:38], c.get('error') or (c['counts'], c['mts'], c['n_v'])))

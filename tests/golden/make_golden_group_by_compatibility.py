"""states.group_by_compatibility executed from the reference's source on stand-in states (states.py:186-217: the function
only calls `state.is_state_compatible(other)`): tests/golden/group_by_compatibility_reference.json holds, for lists of state 'kinds',
the groups and original indices it returns.  tests/test_compat_groups.py runs this package's function on the same stand-ins.
usage: python tests/golden/make_golden_group_by_compatibility.py"""
import ast
import json
import os

REF = '/root/reference/openmmtools/states.py'
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'group_by_compatibility_reference.json')
KIND_LISTS = [[0], [0, 0, 0], [0, 1, 0, 1, 2], [2, 1, 0], [0, 0, 1, 1, 0, 2, 2, 1], [3, 3, 3, 0, 3], []]


class State:
    def __init__(self, kind, index):
        self.kind, self.index = kind, index

    def is_state_compatible(self, other):
        return self.kind == other.kind


def main():
    tree = ast.parse(open(REF).read())
    fn = next(n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == 'group_by_compatibility')
    ns = {}
    exec(compile(ast.Module(body=[fn], type_ignores=[]), REF, 'exec'), ns)
    out = dict(source='openmmtools/states.py:%d-%d' % (fn.lineno, fn.end_lineno), cases=[])
    for kinds in KIND_LISTS:
        states = [State(k, i) for i, k in enumerate(kinds)]
        groups, indices = ns['group_by_compatibility'](states)
        out['cases'].append(dict(kinds=kinds, groups=[[s.index for s in g] for g in groups], original_indices=indices))
    with open(OUT, 'w') as fh:
        json.dump(out, fh)
    print(out)


if __name__ == '__main__':
    main()

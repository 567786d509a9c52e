"""SAMSSampler's option validators executed from the reference's source (sams.py:237-278: static methods of the nested _StoredProperty
class): for every option, what passes and the exact ValueError text of what does not.  tests/golden/sams_validators_reference.json;
tests/test_sampler_cpu.py holds this package's constructor to it.     usage: python tests/golden/make_golden_sams_validators.py"""
import ast
import json
import os

REF = '/root/reference/openmmtools/multistate/sams.py'
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'sams_validators_reference.json')
PROBES = {'state_update_scheme': ['global-jump', 'local-jump', 'restricted-range-jump', 'bogus'],
          'update_stages': ['one-stage', 'two-stage', 'three-stage'],
          'flatness_criteria': ['minimum-visits', 'logZ-flatness', 'histogram-flatness', 'flat'],
          'weight_update_method': ['optimal', 'rao-blackwellized', 'naive'],
          'adapt_target_probabilities': [False, True]}


def main():
    tree = ast.parse(open(REF).read())
    cls = next(n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == 'SAMSSampler')
    sp = next(n for n in cls.body if isinstance(n, ast.ClassDef) and n.name == '_StoredProperty')
    sp.bases = []
    ns = {}
    exec(compile(ast.Module(body=[sp], type_ignores=[]), REF, 'exec'), ns)
    SP = ns['_StoredProperty']
    out = dict(source='openmmtools/multistate/sams.py:%d-%d' % (sp.lineno, sp.end_lineno), options={})
    for option, values in PROBES.items():
        fn = getattr(SP, '_%s_validator' % option)
        rows = []
        for v in values:
            try:
                rows.append(dict(value=v, returns=fn(None, v)))
            except ValueError as exc:
                rows.append(dict(value=v, error=str(exc)))
        out['options'][option] = rows
    with open(OUT, 'w') as fh:
        json.dump(out, fh, indent=1)
    print(json.dumps(out['options'], indent=0)[:900])


if __name__ == '__main__':
    main()

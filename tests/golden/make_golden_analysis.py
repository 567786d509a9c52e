"""Reference-executed vectors for the pure-numpy / pure-Python helpers of the analysis layer:
    multistateanalyzer.py:993-1040  PhaseAnalyzer.reformat_energies_for_mbar
    multistate/utils.py:60-95       generate_phase_name
    multistatesampler.py:1117-1143  MultiStateSampler._default_initial_thermodynamic_states
taken from the syntax trees of the files under /root/reference (decorators and annotations dropped), run here.
Output: tests/golden/analysis_reference.json.     usage: python tests/golden/make_golden_analysis.py"""
import ast
import json
import os
import numpy as np

REF = '/root/reference/openmmtools/multistate'
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'analysis_reference.json')


def take(path, name, cls=None):
    tree = ast.parse(open(path).read())
    body = tree.body if cls is None else next(n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == cls).body
    node = next(n for n in body if isinstance(n, ast.FunctionDef) and n.name == name)
    node.decorator_list = []
    node.returns = None
    for a in node.args.args:
        a.annotation = None
    ns = {'np': np}
    exec(compile(ast.Module(body=[node], type_ignores=[]), path, 'exec'), ns)
    return ns[name]


if __name__ == '__main__':
    reformat = take(os.path.join(REF, 'multistateanalyzer.py'), 'reformat_energies_for_mbar', cls='PhaseAnalyzer')
    phase_name = take(os.path.join(REF, 'utils.py'), 'generate_phase_name')
    assign = take(os.path.join(REF, 'multistatesampler.py'), '_default_initial_thermodynamic_states', cls='MultiStateSampler')
    rng = np.random.default_rng(11)
    u = rng.normal(size=(3, 4, 5))
    cases = dict(
        u_kln=u.tolist(),
        full=reformat(u).tolist(),
        ragged_n_k=[5, 2, 0],
        ragged=reformat(u, np.array([5, 2, 0])).tolist(),
        initial_states=[[k, r, [int(i) for i in assign(None, list(range(k)), list(range(r)))]] for k in range(1, 8) for r in range(1, 10)],
        names=[[cur, lst, phase_name(cur, lst)] for cur, lst in (
            (None, []), (None, ['phase0']), (None, ['phase0', 'phase1', 'x']), ('complex', []), ('complex', ['complex']),
            ('complex', ['complex', 'complex0', 'complex1']), ('solvent', ['complex']))])
    with open(OUT, 'w') as fh:
        json.dump(dict(source='multistateanalyzer.py:993-1040, multistate/utils.py:60-95 executed from /root/reference', **cases), fh)
    print('written', OUT)

"""MultiStateSampler._compute_energies / _compute_replica_energies / _neighborhood EXECUTED from the reference's source
(multistatesampler.py:1263-1281, 1437-1494) on a stand-in sampler: thermodynamic states are stand-ins with a 'kind' (compatibility group),
the "context" returns prepared numbers u[replica][state] through states.ThermodynamicState.reduced_potential_at_states, and
states.group_by_compatibility is the reference's own function.  What is recorded: which entries of the energy matrix an iteration
refreshes under a locality, what stays from the iteration before, the neighbourhood mask, the unsampled states' columns.
Output: tests/golden/compute_energies_reference.json; tests/test_sampler_cpu.py runs this package's method on an engine stand-in that
returns the same numbers.        usage: python tests/golden/make_golden_compute_energies.py"""
import ast
import json
import os

import numpy as np

REF = '/root/reference/openmmtools/multistate/multistatesampler.py'
REF_STATES = '/root/reference/openmmtools/states.py'
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'compute_energies_reference.json')


def build():
    tree = ast.parse(open(REF).read())
    cls = next(n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == 'MultiStateSampler')
    wanted = [n for n in cls.body if isinstance(n, ast.FunctionDef) and n.name in ('_compute_energies', '_compute_replica_energies', '_neighborhood')]
    for f in wanted:
        f.decorator_list = []
    gtree = ast.parse(open(REF_STATES).read())
    gfn = next(n for n in gtree.body if isinstance(n, ast.FunctionDef) and n.name == 'group_by_compatibility')
    gns = {}
    exec(compile(ast.Module(body=[gfn], type_ignores=[]), REF_STATES, 'exec'), gns)

    class Context:
        sampler = None

    class ThermodynamicState:
        @staticmethod
        def reduced_potential_at_states(context, group):
            return [context.sampler.u[s.index] for s in group]

    class states:
        group_by_compatibility = staticmethod(gns['group_by_compatibility'])
    states.ThermodynamicState = ThermodynamicState

    class mpiplus:
        @staticmethod
        def distribute(fn, it, send_results_to=None):
            it = list(it)
            return [fn(i) for i in it], it

    ns = dict(np=np, states=states, mpiplus=mpiplus, logger=None)
    body = ast.ClassDef(name='Sampler', bases=[], keywords=[], body=wanted, decorator_list=[])
    exec(compile(ast.fix_missing_locations(ast.Module(body=[body], type_ignores=[])), REF, 'exec'), ns)
    return ns['Sampler'], Context


class State:
    def __init__(self, kind, index):
        self.kind, self.index = kind, index

    def is_state_compatible(self, other):
        return self.kind == other.kind


class SamplerState:
    def __init__(self, u):
        self.u = u

    def apply_to_context(self, context, ignore_velocities=False):
        context.sampler = self


def main():
    Sampler, Context = build()
    rng = np.random.default_rng(20260927)
    out = dict(cases=[])
    for K, R, U, locality, kinds in ((6, 4, 0, None, [0] * 6), (6, 4, 2, 1, [0] * 8), (7, 5, 1, 2, [0, 0, 1, 1, 0, 2, 2, 0]), (5, 5, 0, 3, [0] * 5)):
        s = Sampler()
        s.locality = locality
        s._thermodynamic_states = [State(kinds[k], k) for k in range(K)]
        s._unsampled_states = [State(kinds[K + k], K + k) for k in range(U)]
        s.n_states, s.n_replicas = K, R
        s._neighborhoods = np.zeros((R, K), dtype=bool)
        s._energy_thermodynamic_states = np.zeros((R, K))
        s._energy_unsampled_states = np.zeros((R, U))

        class Cache:
            def get_context(self, state):
                return Context(), None
        s.energy_context_cache = Cache()
        calls = []
        for it in range(3):
            labels = rng.integers(0, K, R)
            full = rng.normal(size=(R, K + U)).round(6)
            s._replica_thermodynamic_states = labels
            s._sampler_states = [SamplerState(full[r]) for r in range(R)]
            s._compute_energies()
            calls.append(dict(labels=labels.tolist(), full=full.tolist(), energy_thermodynamic_states=s._energy_thermodynamic_states.tolist(),
                              energy_unsampled_states=s._energy_unsampled_states.tolist(), neighborhoods=s._neighborhoods.astype(int).tolist()))
        out['cases'].append(dict(n_states=K, n_replicas=R, n_unsampled=U, locality=locality, kinds=kinds, calls=calls))
        print(K, R, U, locality, 'refreshed entries per call:', [int(np.sum(c['neighborhoods'])) for c in calls])
    with open(OUT, 'w') as fh:
        json.dump(out, fh, separators=(',', ':'))


if __name__ == '__main__':
    main()

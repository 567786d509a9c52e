"""MultiStateSampler.run / extend / _is_completed / _update_timing EXECUTED from the reference's source on a stand-in
sampler whose steps only log their names (multistatesampler.py:724-821, 1717-1739, 1766-1803): the order in which an iteration calls
_mix_replicas / _propagate_replicas / _compute_energies / _report_iteration / _update_analysis, what iteration 0 does first, how run(n) and
extend(n) bound the iterations, when the online-analysis error target ends a run, and the keys of _timing_data.
Output: tests/golden/run_loop_reference.json; tests/test_sampler_cpu.py runs this package's sampler with the same steps logged.
usage: python tests/golden/make_golden_run_loop.py"""
import ast
import datetime
import json
import os

REF = '/root/reference/openmmtools/multistate/multistatesampler.py'
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'run_loop_reference.json')
STEPS = ('_compute_energies', '_check_nan_energy', '_mix_replicas', '_propagate_replicas', '_report_iteration', '_update_analysis')


class _Timer:
    def start(self, name): pass
    def stop(self, name): return 0.25
    def partial(self, name): return 1.5
    def report_timing(self): pass


class _Quantity:
    def __init__(self, v): self.v = v
    def value_in_unit(self, u): return self.v / u.v
    def __rmul__(self, o): return _Quantity(o * self.v)


class _unit:
    nanosecond, seconds, day = _Quantity(1e-9), _Quantity(1.0), _Quantity(86400.0)


class _Move:
    timestep, n_steps = _Quantity(2e-15), 500


def build():
    tree = ast.parse(open(REF).read())
    cls = next(n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == 'MultiStateSampler')
    wanted = [n for n in cls.body if isinstance(n, ast.FunctionDef) and n.name in ('run', 'extend', '_is_completed', '_is_completed_static', '_update_timing')]
    for f in wanted:
        f.decorator_list = [d for d in f.decorator_list if isinstance(d, ast.Name) and d.id == 'staticmethod']
    log = []

    class Reporter:
        def write_energies(self, *a): log.append('reporter.write_energies')

    class mpiplus:
        @staticmethod
        def run_single_node(rank, fn, *a, **kw): return fn(*a, **kw)

    class utils:
        Timer = _Timer

    ns = dict(mpiplus=mpiplus, utils=utils, logger=type('L', (), dict(info=lambda *a, **k: None, debug=lambda *a, **k: None, critical=lambda *a, **k: None))(),
              datetime=datetime, unit=_unit, SimulationNaNError=RuntimeError)
    body = ast.ClassDef(name='Sampler', bases=[], keywords=[], body=wanted, decorator_list=[])
    exec(compile(ast.fix_missing_locations(ast.Module(body=[body], type_ignores=[])), REF, 'exec'), ns)
    Sampler = ns['Sampler']

    def make(iteration, number_of_iterations, target=0.0, errors=()):
        s = Sampler()
        s._iteration, s.number_of_iterations = iteration, number_of_iterations
        s._reporter, s._timing_data = Reporter(), {}
        s._energy_thermodynamic_states = s._neighborhoods = s._energy_unsampled_states = None
        s.online_analysis_target_error = target
        s._last_err_free_energy = None
        errs = list(errors)
        for name in STEPS:
            def step(name=name):
                log.append(name)
                if name == '_update_analysis' and errs:
                    s._last_err_free_energy = errs.pop(0)
                return 'labels' if name == '_mix_replicas' else None
            setattr(s, name, step)
        s._flatten_moves_iterator = lambda: [_Move(), _Move()]
        return s
    return make, log, {f.name: (f.lineno, f.end_lineno) for f in wanted}


def main():
    make, log, lines = build()
    out = dict(source={k: 'multistatesampler.py:%d-%d' % v for k, v in lines.items()}, cases=[])

    def case(label, s, call):
        del log[:]
        call(s)
        out['cases'].append(dict(label=label, calls=list(log), iteration=s._iteration, number_of_iterations=s.number_of_iterations,
                                 timing_keys=sorted(s._timing_data)))
    case('run(2) from iteration 0 of 5', make(0, 5), lambda s: s.run(2))
    case('run() from iteration 3 of 5', make(3, 5), lambda s: s.run())
    case('run(10) from iteration 4 of 5', make(4, 5), lambda s: s.run(10))
    case('extend(2) at iteration 5 of 5', make(5, 5), lambda s: s.extend(2))
    case('run() of 6 with an error target reached after the 2nd analysis', make(0, 6, target=0.5, errors=[0.9, 0.4, 0.1]), lambda s: s.run())
    s = make(2, 10)
    s._iteration = 4
    s._update_timing(0.25, 1.5, 2, 10)
    out['timing_example'] = dict(iteration=4, run_initial_iteration=2, iteration_limit=10, iteration_time=0.25, partial_total_time=1.5,
                                 moves=[dict(timestep_ps=2e-3, n_steps=500)] * 2,
                                 timing_data={k: v for k, v in s._timing_data.items() if k != 'estimated_localtime_finish_date'})
    with open(OUT, 'w') as fh:
        json.dump(out, fh, indent=1)
    for c in out['cases']:
        print(c['label'], '->', c['iteration'], c['calls'])
    print(out['timing_example']['timing_data'])


if __name__ == '__main__':
    main()

"""Same-box A/B of variant builds of libremd_hip (tools/build_variant.sh, or hand-built openmmtools_amd/libremd_hip_<name>.so):
per library the stand-alone time per launch of the direct-space kernels, the forces compared bit for bit with the first library's,
and ms per 500 MD steps of the whole step, interleaved over `rounds` rounds so that clock drift of the box hits all alike.
usage: python tools/ab_libs.py [--R 24] [--system alanine] [--rounds 2] [--steps 500] name1 name2 ...   ('tree' = libremd_hip.so)"""
import os, sys, argparse
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from openmmtools_amd import testsystems as ts
from openmmtools_amd.system import system_to_desc
from openmmtools_amd._engine import HipEngine
KB = 0.008314462618153242
ap = argparse.ArgumentParser()
ap.add_argument('--R', type=int, default=24)
ap.add_argument('--system', default='alanine')
ap.add_argument('--rounds', type=int, default=2)
ap.add_argument('--steps', type=int, default=500)
ap.add_argument('--split', default='auto')
ap.add_argument('--no-insitu', action='store_true')
ap.add_argument('names', nargs='+')
a = ap.parse_args()
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
libs = {n: (None if n == 'tree' else os.path.join(root, 'openmmtools_amd', 'libremd_hip_%s.so' % n)) for n in a.names}
al = {'alanine': ts.AlanineDipeptideExplicit, 'hostguest': ts.HostGuestExplicit, 'dhfr': ts.DHFRExplicit}[a.system]()
box = np.diag(al.system.getDefaultPeriodicBoxVectors())
d = system_to_desc(al.system, ewald_split=a.split if a.split in ('auto', 'reference') else float(a.split))
R = a.R
# jiggled start (identical replicas would make every replica's list identical)
rng = np.random.default_rng(7)
x0 = np.tile(al.positions, (R, 1, 1)) + rng.normal(0, 0.002, (R,) + al.positions.shape)


def engine(name, overlap):
    os.environ['REMD_OVERLAP'] = '1' if overlap else '0'
    eng = HipEngine(lib_path=libs[name], ewald_split=a.split)
    eng.set_system(d); eng.set_states(1 / (KB * np.geomspace(300.0, 600.0, R)))
    return eng


ref = None
res = {n: {'sa': [], 'sa_all': [], 'ms': []} for n in a.names}
for rnd in range(a.rounds):
    for n in a.names:
        eng = engine(n, False)
        eng.set_integrator('V R O R V', 0.001, 1.0, 1, True, 1e-8)
        eng.set_replicas(R, 0, x0, None, np.tile(box, (R, 1)), np.arange(R))
        f = eng.get_forces()
        if ref is None:
            ref = f.copy()
        if rnd == 0:
            res[n]['dF'] = float(np.abs(f - ref).max()); res[n]['bit'] = bool(np.array_equal(f, ref))
        eng.profile_enable(2); eng.profile_reset()
        for _ in range(30):
            eng.get_forces()
        out = {k: eng.profile_get(k) for k in ('nonbonded', 'nb_gather', 'pme_fft', 'bonded')}
        us = {k: 1e3 * v[1] / max(1, v[0]) for k, v in out.items()}
        res[n]['sa'].append(us['nonbonded']); res[n]['sa_all'].append(us)
        eng.close()
    if a.no_insitu:
        continue
    for n in a.names:
        eng = engine(n, True)
        n_steps = a.steps
        eng.set_integrator('V R R O R R V', 0.002, 1.0, n_steps, True, 1e-8)
        eng.set_replicas(R, 0, x0, None, np.tile(box, (R, 1)), np.arange(R))
        eng.propagate(0)
        ms = []
        for it in range(1, 5):
            eng.propagate(it); ms.append(eng.last_timing()['propagate_ms'] * 500.0 / n_steps)
        res[n]['ms'].append(min(ms))
        if rnd == 0:
            xr = eng.get_replicas()[0]
            res[n]['xsum'] = float(np.abs(xr).sum())
        eng.close()
print('%s x %d  split %s' % (a.system, R, a.split))
for n in a.names:
    r = res[n]
    print('%-12s pair+scatter stand-alone us: %s   gather+list %s | ms/500 steps: %s | forces bit-identical to %s: %s (max |dF| %.3g)  xsum %.9g' % (
        n, ' '.join('%.1f' % v for v in r['sa']), ' '.join('%.1f' % u['nb_gather'] for u in r['sa_all']),
        ' '.join('%.2f' % v for v in r['ms']), a.names[0], r.get('bit'), r.get('dF', -1), r.get('xsum', 0.0)), flush=True)

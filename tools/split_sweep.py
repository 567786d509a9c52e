"""ms per 500 MD steps of R alanine-dipeptide replicas for one Ewald split; environment variants
(REMD_NB_TABLE, REMD_CU_PAIR / REMD_CU_MESH / REMD_CU_LAYOUT, REMD_NB_PERSIST_GRID) are per process.
usage: python tools/split_sweep.py <reference|auto|r_coul_nm> [R] [system: alanine|hostguest|dhfr] [standalone]"""
import os, sys, time
if len(sys.argv) > 4 and sys.argv[4] == 'standalone':
    os.environ['REMD_OVERLAP'] = '0'
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from openmmtools_amd import testsystems as ts
from openmmtools_amd.system import system_to_desc
from openmmtools_amd._engine import HipEngine
KB = 0.008314462618153242
split = sys.argv[1] if len(sys.argv) > 1 else 'reference'
if split not in ('reference', 'auto'):
    split = float(split)
R = int(sys.argv[2]) if len(sys.argv) > 2 else 24
name = sys.argv[3] if len(sys.argv) > 3 else 'alanine'
standalone = len(sys.argv) > 4 and sys.argv[4] == 'standalone'
al = {'alanine': ts.AlanineDipeptideExplicit, 'hostguest': ts.HostGuestExplicit, 'dhfr': ts.DHFRExplicit}[name]()
box = np.diag(al.system.getDefaultPeriodicBoxVectors())
d = system_to_desc(al.system, ewald_split=split)
eng = HipEngine(lib_path=os.environ.get('AB_LIB') or None)      # AB_LIB: a variant build (tools/build_variant.sh)
eng.set_system(d); eng.set_states(1 / (KB * np.geomspace(300.0, 600.0, R)))
tag = 'split %-9s rcoul %.3f mesh %s  env {%s}' % (sys.argv[1] if len(sys.argv) > 1 else 'reference', d.get('coulomb_cutoff', d['cutoff']), list(d['pme_grid']),
      ' '.join('%s=%s' % (k[5:], v) for k, v in sorted(os.environ.items()) if k.startswith('REMD_')) + (' lib=' + os.path.basename(os.environ['AB_LIB']) if os.environ.get('AB_LIB') else ''))
if standalone:
    eng.set_integrator('V R O R V', 0.001, 1.0, 1, True, 1e-8)
    eng.set_replicas(R, 0, np.tile(al.positions, (R, 1, 1)), None, np.tile(box, (R, 1)), np.arange(R))
    eng.get_forces()
    eng.profile_enable(2); eng.profile_reset()
    for _ in range(20):
        eng.get_forces()
    out = {k: eng.profile_get(k) for k in ('pme_fft', 'pme_xy', 'pme_zinv_gather', 'pme_bin', 'nonbonded', 'nb_gather', 'bonded')}
    us = {k: round(1e3 * v[1] / max(1, v[0]), 1) for k, v in out.items()}
    us['pme_spread_zfwd'] = round(us['pme_fft'] - us['pme_xy'] - us['pme_zinv_gather'], 1)
    print(tag, 'R', R, 'standalone us per launch', us, flush=True)
else:
    n_steps = 500 if name != 'dhfr' else 100
    eng.set_integrator('V R R O R R V', 0.002, 1.0, n_steps, True, 1e-8)
    eng.set_replicas(R, 0, np.tile(al.positions, (R, 1, 1)), None, np.tile(box, (R, 1)), np.arange(R))
    eng.propagate(0)
    ms = []
    for it in range(1, 5):
        eng.propagate(it); ms.append(eng.last_timing()['propagate_ms'] * 500.0 / n_steps)
    print(tag, 'R', R, 'ms per 500 steps: min %.2f  all %s' % (min(ms), ' '.join('%.1f' % m for m in ms)), flush=True)
eng.close()

"""Does splitting one GPU's replicas over several handles (each with its own streams), driven from host threads, hide
the latency floor of a step?  usage: python tools/multi_engine_check.py <n_engines> [R_total]"""
import os, sys, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from openmmtools_amd import testsystems as ts
from openmmtools_amd.system import system_to_desc
from openmmtools_amd._engine import HipEngine
KB = 0.008314462618153242
al = ts.AlanineDipeptideExplicit()
box = np.diag(al.system.getDefaultPeriodicBoxVectors())
n_eng = int(sys.argv[1]) if len(sys.argv) > 1 else 2
R = int(sys.argv[2]) if len(sys.argv) > 2 else 24
desc = system_to_desc(al.system)
T = np.geomspace(300.0, 600.0, R)
engines, b = [], 0
for e in range(n_eng):
    c = R // n_eng + (1 if e < R % n_eng else 0)
    eng = HipEngine()
    eng.set_system(desc); eng.set_states(1 / (KB * T))
    eng.set_integrator('V R R O R R V', 0.002, 1.0, 500, True, 1e-8)
    eng.seed(0xC0FFEE)
    eng.set_replicas(R, b, np.tile(al.positions, (c, 1, 1)), None, np.tile(box, (c, 1)), np.arange(R))
    engines.append(eng); b += c


def run(it):
    th = [threading.Thread(target=e.propagate, args=(it,)) for e in engines]
    t0 = time.perf_counter()
    for t in th: t.start()
    for t in th: t.join()
    return time.perf_counter() - t0


run(0)
for it in range(1, 4):
    print('engines', n_eng, 'R', R, 'wall ms per 500 steps', 1e3 * run(it))
x = np.concatenate([e.get_replicas()[0] for e in engines])
print('checksum', float(np.abs(x).sum()))

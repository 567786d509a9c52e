"""Experiment: several handles (replica groups) on one GPU driven step by step from ONE host thread, so that the latency-bound
phases of one group overlap the force kernels of another.  usage: python tools/interleave_check.py <n_groups> [R]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from openmmtools_amd import testsystems as ts
from openmmtools_amd.system import system_to_desc
from openmmtools_amd._engine import HipEngine
KB = 0.008314462618153242
al = ts.AlanineDipeptideExplicit()
box = np.diag(al.system.getDefaultPeriodicBoxVectors())
G = int(sys.argv[1]) if len(sys.argv) > 1 else 2
R = int(sys.argv[2]) if len(sys.argv) > 2 else 24
chunk = int(sys.argv[3]) if len(sys.argv) > 3 else 1
desc = system_to_desc(al.system)
T = np.geomspace(300.0, 600.0, R)
engines, b = [], 0
for e in range(G):
    c = R // G + (1 if e < R % G else 0)
    eng = HipEngine()
    eng.set_system(desc); eng.set_states(1 / (KB * T))
    eng.set_integrator('V R R O R R V', 0.002, 1.0, 500, True, 1e-8)
    eng.seed(0xC0FFEE)
    eng.set_replicas(R, b, np.tile(al.positions, (c, 1, 1)), None, np.tile(box, (c, 1)), np.arange(R))
    eng.propagate(0)
    engines.append(eng); b += c
n = 500
for rep in range(3):
    t0 = time.perf_counter()
    for s in range(0, n, chunk):
        for e in engines:
            e.step('V R R O R R V', iteration=1 + rep, first_step=s, n_steps=chunk)
    t1 = time.perf_counter()
    for e in engines: e.sync()
    t2 = time.perf_counter()
    print('groups', G, 'R', R, 'chunk', chunk, 'enqueue ms', 1e3 * (t1 - t0), 'total ms per 500 steps', 1e3 * (t2 - t0))

"""One-GPU measurements from which the multi-GPU curves follow: the mix / propagate / u_kl split of one
iteration for R replicas of the headline system (parallel tempering, swap-all) and of DHFR (SAMS global jump, 128 states) on
ONE GPU.  Sharding moves labels, not coordinates, and the only collective is the all-gather of u_kl rows (<= 128 KiB), so an
N-GPU iteration is   propagate(R / N) + u_kl(R / N) + mix(R, replicated) + all-gather:
    weak  scaling, 24 replicas per GPU:  eff(N) = t_iter(24) / [prop(24) + ukl(24) + mix(24 N)]
    strong scaling, R_tot replicas:      eff(N) = t_iter(R_tot on one GPU) / (N [prop(R_tot / N) + ukl(R_tot / N) + mix(R_tot)])
(the all-gather is a few tens of microseconds over xGMI and is left out; the driver's SCALE run measures the real thing).
usage: python tools/scaling_projection.py [alanine] [dhfr]  -> markdown on stdout"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from openmmtools_amd import testsystems, states, mcmc, unit
from openmmtools_amd.multistate import ParallelTemperingSampler, SAMSSampler
from openmmtools_amd._engine import HipEngine

which = sys.argv[1:] or ['alanine', 'dhfr']


def move(n_steps):
    return mcmc.LangevinSplittingDynamicsMove(timestep=2.0 * unit.femtosecond, collision_rate=1.0 / unit.picosecond, n_steps=n_steps,
                                              reassign_velocities=True, splitting='V R R O R R V')


def timed(s, n_iter, scale):
    s.run(1)
    torch.cuda.synchronize()
    acc = dict(mixing_seconds=0.0, propagation_seconds=0.0, energy_seconds=0.0)
    t0 = time.perf_counter()
    for _ in range(n_iter):
        s.run(1)
        for k in acc:
            acc[k] += float(s._timing_data[k])
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / n_iter
    mix, prop, en = [1e3 * acc[k] / n_iter for k in ('mixing_seconds', 'propagation_seconds', 'energy_seconds')]
    return mix, prop * scale, en, 1e3 * wall + prop * (scale - 1.0)


rows = {}
if 'alanine' in which:
    al = testsystems.AlanineDipeptideExplicit()
    for R in (8, 12, 16, 24, 32, 48, 64, 96, 128, 192):
        eng = HipEngine()
        s = ParallelTemperingSampler(mcmc_moves=move(500), number_of_iterations=10 ** 9, engine=eng, seed=0xC0FFEE)
        s.create(states.ThermodynamicState(al.system, 300.0 * unit.kelvin),
                 [states.SamplerState(al.positions, box_vectors=al.system.getDefaultPeriodicBoxVectors())], storage=None,
                 min_temperature=300.0 * unit.kelvin, max_temperature=600.0 * unit.kelvin, n_temperatures=R)
        rows[('alanine', R)] = timed(s, 2, 1.0)
        eng.close()
    print('### AlanineDipeptideExplicit, parallel tempering logspace(300 K, 600 K), swap-all, 500 MD steps per iteration, one GPU\n')
    print('| replicas on the GPU | mix ms | propagate ms | u_kl ms | iteration ms |\n|---|---|---|---|---|')
    for (name, R), v in rows.items():
        if name == 'alanine':
            print('| %d | %.2f | %.1f | %.2f | %.1f |' % ((R,) + v))
    a = {R: v for (n, R), v in rows.items() if n == 'alanine'}
    print('\nweak scaling (24 replicas per GPU; mix of the 24 N-replica ensemble replicated on every rank):\n')
    print('| GPUs | replicas | prop + u_kl (24) ms | mix(24 N) ms | iteration ms | efficiency |\n|---|---|---|---|---|---|')
    t1 = a[24][1] + a[24][2] + a[24][0]
    for N in (1, 2, 4, 8):
        tn = a[24][1] + a[24][2] + a[24 * N][0]
        print('| %d | %d | %.1f | %.2f | %.1f | %.3f |' % (N, 24 * N, a[24][1] + a[24][2], a[24 * N][0], tn, t1 / tn))
    for Rt in (24, 128):
        print('\nstrong scaling, one %d-replica ensemble:\n' % Rt)
        print('| GPUs | replicas per GPU | prop + u_kl ms | mix(%d) ms | iteration ms | speed-up | efficiency |\n|---|---|---|---|---|---|---|' % Rt)
        t1 = a[Rt][1] + a[Rt][2] + a[Rt][0]
        for N in (1, 2, 4, 8):
            per = -(-Rt // N)
            if per not in a:
                continue
            tn = a[per][1] + a[per][2] + a[Rt][0]
            print('| %d | %d | %.1f | %.2f | %.1f | %.2f | %.3f |' % (N, per, a[per][1] + a[per][2], a[Rt][0], tn, t1 / tn, t1 / tn / N))
if 'dhfr' in which:
    dh = testsystems.DHFRExplicit()
    T = np.geomspace(300.0, 400.0, 128)
    ths = [states.ThermodynamicState(dh.system, t) for t in T]
    ss = states.SamplerState(dh.positions, box_vectors=dh.system.getDefaultPeriodicBoxVectors())
    MD = 100                                         # of 500: propagation scaled x5 (it is linear in the step count)
    for R in (16, 32, 64, 128):
        eng = HipEngine()
        s = SAMSSampler(mcmc_moves=move(MD), number_of_iterations=10 ** 9, engine=eng, seed=1)
        s.create(ths, [ss] * R)
        rows[('dhfr', R)] = timed(s, 1, 500.0 / MD)
        eng.close()
    print('\n### DHFRExplicit (23 558 atoms), 128 temperature states, SAMS global jump, 500 MD steps per iteration '
          '(measured with %d, propagation scaled), one GPU\n' % MD)
    print('| replicas on the GPU | mix ms | propagate ms | u_kl ms | iteration ms |\n|---|---|---|---|---|')
    d = {R: v for (n, R), v in rows.items() if n == 'dhfr'}
    for R, v in d.items():
        print('| %d | %.2f | %.1f | %.2f | %.1f |' % ((R,) + v))
    print('\nstrong scaling of the 128-replica ensemble (config 5 / north_star shape):\n')
    print('| GPUs | replicas per GPU | prop + u_kl ms | mix ms | iteration ms | speed-up | efficiency |\n|---|---|---|---|---|---|---|')
    t1 = sum(d[128][:3])
    for N in (1, 2, 4, 8):
        per = 128 // N
        tn = d[per][1] + d[per][2] + d[128][0]
        print('| %d | %d | %.1f | %.2f | %.1f | %.2f | %.3f |' % (N, per, d[per][1] + d[per][2], d[128][0], tn, t1 / tn, t1 / tn / N))

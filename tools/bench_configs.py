"""Iterations/s of the single-GPU share of the other BASELINE configs, same engine and protocol as
bench.py: config 2 (LJ fluid 512, 16 lambda_sterics states), config 4 share (CB7:B2 host-guest, 8 replicas per GPU out
of 64 alchemical states), config 5 share (DHFR, 16 replicas per GPU out of 128 states, SAMS global jump).
Prints one JSON line per config.  (The headline config 3 is bench.py.)"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from openmmtools_amd import testsystems, states, mcmc, unit, alchemy
from openmmtools_amd.multistate import ReplicaExchangeSampler, SAMSSampler, ParallelTemperingSampler
from openmmtools_amd._engine import HipEngine


def alchemical_states(system, atoms, lam_e, lam_s, T=300.0):
    region = alchemy.AlchemicalRegion(alchemical_atoms=atoms)
    asys = alchemy.AbsoluteAlchemicalFactory().create_alchemical_system(system, region)
    out = []
    for le, ls in zip(lam_e, lam_s):
        a = states.AlchemicalState(lambda_sterics=ls, lambda_electrostatics=le)
        out.append(states.CompoundThermodynamicState(states.ThermodynamicState(asys, T), [a]))
    return out


def run(name, sampler, n_iter, warm=1):
    sampler.run(warm)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    sampler.run(n_iter)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n_iter
    print(json.dumps(dict(config=name, iterations_per_s=1.0 / dt, ms_per_iteration=1e3 * dt, replicas=sampler.n_replicas,
                          states=sampler.n_states, timing={k: (v if isinstance(v, str) else float(v)) for k, v in sampler._timing_data.items()})), flush=True)


def main():
    which = sys.argv[1:] or ['2', '4', '5', '5h']
    def move(dt_fs, split):
        return mcmc.LangevinSplittingDynamicsMove(timestep=dt_fs * unit.femtosecond, collision_rate=1.0 / unit.picosecond,
                                                  n_steps=500, reassign_velocities=True, splitting=split)
    if '2' in which:
        lj = testsystems.LennardJonesFluid(nparticles=512)
        ths = alchemical_states(lj.system, range(10), np.ones(16), np.linspace(1.0, 0.0, 16))
        s = ReplicaExchangeSampler(mcmc_moves=move(1.0, 'V R O R V'), number_of_iterations=10 ** 9, engine=HipEngine(), seed=1)
        s.create(ths, [states.SamplerState(lj.positions, box_vectors=lj.system.getDefaultPeriodicBoxVectors())])
        run('2: LennardJonesFluid(512), 16 lambda_sterics states, BAOAB 1 fs x 500', s, 5)
    if '4' in which:
        hg = testsystems.HostGuestExplicit()
        lam_e = np.concatenate([np.linspace(1.0, 0.0, 32), np.zeros(32)])
        lam_s = np.concatenate([np.ones(32), np.linspace(1.0, 0.0, 32)])
        ths = alchemical_states(hg.system, range(126, 156), lam_e, lam_s)
        # one GPU's share of the 64-replica ensemble: 8 replicas, all 64 states in u_kl (R != K => no swap-all; SAMS jump)
        s = SAMSSampler(mcmc_moves=move(2.0, 'V R R O R R V'), number_of_iterations=10 ** 9, engine=HipEngine(), seed=1)
        ss = states.SamplerState(hg.positions, box_vectors=hg.system.getDefaultPeriodicBoxVectors())
        s.create(ths, [ss] * 8)
        run('4 (1-GPU share): HostGuestExplicit, 8 replicas x 64 alchemical states, g-BAOAB 2 fs x 500', s, 3)
    for tag, kw in (('4r', dict()), ('4d', dict(alchemical_pme_treatment='direct-space'))):
        if tag not in which:
            continue
        # config 4's share on the GENERAL alchemical path (csrc/alch_regions.hip): the guest and the reference's second test region
        # (atoms 156-159, tests/test_alchemy.py:2203-2208) as two named regions, both on config 4's ladder; '4r': the exact PME treatment
        # (the regions' scaled charges inside the Ewald sum, u_kl from six energy passes), '4d': 'direct-space' soft-core electrostatics
        hg = testsystems.HostGuestExplicit()
        lam_e = np.concatenate([np.linspace(1.0, 0.0, 32), np.zeros(32)])
        lam_s = np.concatenate([np.ones(32), np.linspace(1.0, 0.0, 32)])
        regions = [alchemy.AlchemicalRegion(alchemical_atoms=range(126, 156), name='zero'), alchemy.AlchemicalRegion(alchemical_atoms=range(156, 160), name='one')]
        asys = alchemy.AbsoluteAlchemicalFactory(**kw).create_alchemical_system(hg.system, regions)
        ths = [states.CompoundThermodynamicState(states.ThermodynamicState(asys, 300.0),
                                                 [states.AlchemicalState(parameters_name_suffix='zero', lambda_sterics=ls, lambda_electrostatics=le),
                                                  states.AlchemicalState(parameters_name_suffix='one', lambda_sterics=ls, lambda_electrostatics=le)])
               for le, ls in zip(lam_e, lam_s)]
        s = SAMSSampler(mcmc_moves=move(2.0, 'V R R O R R V'), number_of_iterations=10 ** 9, engine=HipEngine(), seed=1)
        ss = states.SamplerState(hg.positions, box_vectors=hg.system.getDefaultPeriodicBoxVectors())
        s.create(ths, [ss] * 8)
        run('%s (1-GPU share, general alchemical regions, %s): HostGuestExplicit, 8 replicas x 64 states of two named regions, g-BAOAB 2 fs x 500'
            % (tag, kw.get('alchemical_pme_treatment', 'exact PME')), s, 3)
    for tag, cls in (('v', 'AlanineDipeptideVacuum'), ('g', 'AlanineDipeptideImplicit'), ('hv', 'HostGuestVacuum')):
        if tag not in which:
            continue
        # the reference's small NoCutoff test systems (csrc/nocutoff.hip, csrc/gbsa.hip): 24 temperatures, the headline's protocol
        t = getattr(testsystems, cls)()
        nt = int(os.environ.get('REMD_BENCH_NT', '24'))           # (swap-all timings at other ensemble sizes: profiles/r06_44)
        ths = [states.ThermodynamicState(t.system, T) for T in np.geomspace(300.0, 600.0, nt)]
        s = ReplicaExchangeSampler(mcmc_moves=move(2.0, 'V R R O R R V'), number_of_iterations=10 ** 9, engine=HipEngine(), seed=1)
        s.create(ths, [states.SamplerState(t.positions)])
        run('%s: %s (%d atoms, NoCutoff), %d temperatures, swap-all, g-BAOAB 2 fs x 500' % (tag, cls, t.system.getNumParticles(), nt), s, 5)
    if 'n' in which:
        # the headline ensemble at constant pressure (NPT states: Monte Carlo barostat every 25 steps inside the propagation)
        al = testsystems.AlanineDipeptideExplicit()
        ths = [states.ThermodynamicState(al.system, T, pressure=1.0 * unit.atmosphere) for T in np.geomspace(300.0, 600.0, 24)]
        s = ReplicaExchangeSampler(mcmc_moves=move(2.0, 'V R R O R R V'), number_of_iterations=10 ** 9, engine=HipEngine(), seed=1)
        s.create(ths, [states.SamplerState(al.positions, box_vectors=al.system.getDefaultPeriodicBoxVectors())])
        run('n: AlanineDipeptideExplicit, 24 temperatures at 1 atm (Monte Carlo barostat every 25 steps), swap-all, g-BAOAB 2 fs x 500', s, 5)
    if '5' in which:
        dh = testsystems.DHFRExplicit()
        T = np.geomspace(300.0, 400.0, 128)
        ths = [states.ThermodynamicState(dh.system, t) for t in T]
        s = SAMSSampler(mcmc_moves=move(2.0, 'V R R O R R V'), number_of_iterations=10 ** 9, engine=HipEngine(), seed=1)
        ss = states.SamplerState(dh.positions, box_vectors=dh.system.getDefaultPeriodicBoxVectors())
        s.create(ths, [ss] * 16)
        run('5 (1-GPU share): DHFRExplicit 23558 atoms, 16 replicas x 128 temperature states, SAMS global jump, g-BAOAB 2 fs x 500', s, 2)
    if '5h' in which:
        # the same share with north_star's wording: 128 HAMILTONIAN replicas = an alchemical ladder (ten solvent molecules decoupled)
        dh = testsystems.DHFRExplicit()
        n = dh.system.getNumParticles()
        lam_e = np.concatenate([np.linspace(1.0, 0.0, 64), np.zeros(64)])
        lam_s = np.concatenate([np.ones(64), np.linspace(1.0, 0.0, 64)])
        ths = alchemical_states(dh.system, range(n - 30, n), lam_e, lam_s)
        s = SAMSSampler(mcmc_moves=move(2.0, 'V R R O R R V'), number_of_iterations=10 ** 9, engine=HipEngine(), seed=1)
        ss = states.SamplerState(dh.positions, box_vectors=dh.system.getDefaultPeriodicBoxVectors())
        s.create(ths, [ss] * 16)
        run('5h (1-GPU share): DHFRExplicit 23558 atoms, 16 replicas x 128 alchemical states, SAMS global jump, g-BAOAB 2 fs x 500', s, 2)


if __name__ == '__main__':
    main()

#!/usr/bin/env python
"""bench.py — REMD iterations/s on the BASELINE.json headline workload.

A "step" is one replica-exchange iteration (mix -> propagate -> u_kl, the loop body of
openmmtools/multistate/multistatesampler.py:766-804 with storage/analysis excluded) of
ParallelTemperingSampler on testsystems.AlanineDipeptideExplicit: 24 replicas per GPU,
temperatures logspace(300 K, 600 K), g-BAOAB "V R R O R R V", 2 fs, 500 MD steps per iteration,
velocities reassigned each iteration, Philox seed 0xC0FFEE.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Default (weak scaling): every rank owns 24 replicas of one 24*N-replica ensemble; the only data-path
collective is the RCCL all-gather of u_kl rows.  `value` counts 24-replica-iteration units:
(R_total / 24) * iterations / s, so that N = 1 is exactly the BASELINE metric.

    --replicas-total T    strong scaling: ONE T-replica ensemble block-sharded over the N ranks
                          (T = 24: the BASELINE metric as written, "24-replica ..., 1/2/4/8 GPU";
                           T = 128: the north_star efficiency target, "1->8 GPUs at 128 replicas").
                          `value` keeps the same unit (T / 24 * iterations / s), `scaling` = "strong".

shapes: besides the main line's `value`, the same JSON line carries `"shapes"`: the other ensemble shapes the scaling question is
asked on, each measured in this very run on the same N ranks (a few iterations each; `--no-shapes` skips them):
    strong24_alanine       one 24-replica ensemble over the N GPUs  (BASELINE.json's metric as written; = the main line at N = 1)
    strong128_alanine      one 128-replica alanine ensemble over the N GPUs (north_star's "1 -> 8 GPUs at 128 replicas")
    config5_dhfr128_sams   DHFR x 128 temperature states, SAMS global jump, over the N GPUs (BASELINE config 5: mixing is R x K)
each with its own value, ms per iteration, the slowest rank's mix / propagate / u_kl / all-gather ms and the world size it saw.

cpu_baseline: the same iteration through the same C ABI on oracle/_build/libremd_cpu.so (f64, OpenMP over
replicas, all host cores), timed on a bounded sample (fewer MD steps per iteration, scaled to 500).
"""
import argparse
import json
import os
import sys
import time

# (before torch / HIP start: two hardware queues per stream priority keep the process at the chip's four pipes when the engine runs its
# replicas as two phases -- openmmtools_amd/__init__.py, include/remd_hip.h: remd_set_phases)
os.environ.setdefault('GPU_MAX_HW_QUEUES', '2')

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

REPLICAS_PER_GPU = 24
MD_STEPS = 500
SEED = 0xC0FFEE
# algorithmic work of the direct-space nonbonded kernel: ~210 pairs/atom inside the
# 1 nm cutoff x ~48 flop/pair (LJ + switch + erfc Coulomb)  =>  10 kflop / atom / force evaluation
FLOP_PER_ATOM_NONBONDED = 1.0e4
FP32_PEAK_TFLOPS = 157.3          # MI355X_MICROARCH.md: FP32 vector peak = f32-input MFMA peak
HBM_PEAK_GBS = 8000.0             # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def pmc_traffic_bytes(kernel):
    """HBM-side bytes per launch of `kernel`, measured offline with `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE`
    (separate passes, FETCH doubled per MI355X_MICROARCH.md) on this same workload; profiles/README.md has the recipe.
    None when no measurement is committed."""
    path = os.path.join(ROOT, 'profiles', 'pmc_traffic_latest.json')
    try:
        with open(path) as fh:
            table = json.load(fh)
        # one force evaluation launches the Coulomb and the LJ sub-system instantiation (force-only, non-alchemical)
        # (force-only, non-alchemical instantiations: the pair kernel's '<3, 0, false, false, ...>', the plane pass's '<64, 8, false>')
        cand = {k: v for k, v in table.items() if kernel in k}
        hits = [v for k, v in cand.items() if '<' not in k or ', false, false' in k]
        if not hits:
            hits = [v for k, v in cand.items() if k.rstrip().endswith(', false>')]
        if hits:
            return sum(float(v['hbm_mb_corrected']) for v in hits) * 1.0e6
    except Exception:
        pass
    return None


def pmc_traffic_is_current():
    """The counters were collected on a build of the pair kernel: profiles/pmc_traffic_latest.source carries the SHA-256 of
    openmmtools_amd/csrc/forces.hip at collection time.  True / False: the file as it is now has / has not that hash (a changed kernel with
    an old table would otherwise go unnoticed); None: no hash recorded."""
    import hashlib, re
    try:
        with open(os.path.join(ROOT, 'profiles', 'pmc_traffic_latest.source')) as fh:
            m = re.search(r'forces\.hip sha256[: ]+([0-9a-f]{64})', fh.read())
        if not m:
            return None
        with open(os.path.join(ROOT, 'openmmtools_amd', 'csrc', 'forces.hip'), 'rb') as fh:
            return hashlib.sha256(fh.read()).hexdigest() == m.group(1)
    except Exception:
        return None


def pmc_traffic_source():
    """Where roofline.traffic comes from: it is NOT measured inside this run (counter passes serialise the kernels); the table
    was collected with tools/collect_profiles.sh on the commit named in profiles/pmc_traffic_latest.source."""
    try:
        with open(os.path.join(ROOT, 'profiles', 'pmc_traffic_latest.source')) as fh:
            return fh.read().strip()
    except Exception:
        return 'profiles/pmc_traffic_latest.json (collection commit not recorded)' 


def build_sampler(n_replicas, engine, comm, md_steps):
    from openmmtools_amd import testsystems, states, mcmc, unit
    from openmmtools_amd.multistate import ParallelTemperingSampler
    ts = testsystems.AlanineDipeptideExplicit()
    thermo = states.ThermodynamicState(ts.system, 300.0 * unit.kelvin)
    sstate = states.SamplerState(ts.positions, box_vectors=ts.system.getDefaultPeriodicBoxVectors())
    move = mcmc.LangevinSplittingDynamicsMove(timestep=2.0 * unit.femtosecond, collision_rate=1.0 / unit.picosecond,
                                              n_steps=md_steps, reassign_velocities=True,
                                              splitting='V R R O R R V')
    sampler = ParallelTemperingSampler(mcmc_moves=move, number_of_iterations=10 ** 9, engine=engine, seed=SEED,
                                       comm=comm)
    sampler.create(thermo, [sstate], storage=None, min_temperature=300.0 * unit.kelvin,
                   max_temperature=600.0 * unit.kelvin, n_temperatures=n_replicas)
    return sampler, ts


def cpu_baseline(n_replicas=REPLICAS_PER_GPU, budget_s=12.0, all_cores=True):
    """The CPU baseline BASELINE.md section 3 names: the same mix -> propagate -> u_kl iteration of the same 24-replica
    parallel-tempering ensemble, through the same C ABI (include/remd_hip.h) implemented on the CPU by
    oracle/_build/libremd_cpu.so (f64, cell/Verlet-list direct space, smooth PME with an in-tree FFT, SETTLE/SHAKE,
    OpenMP over replicas as the reference's mpiplus path distributes replicas over ranks).  Bounded sample: a probe of 2
    MD steps sizes the sample so that the timed iteration costs about `budget_s` seconds on this box; the propagation
    time is scaled linearly to 500 steps, mixing and the energy matrix are timed in full."""
    import oracle
    from openmmtools_amd._engine import HipEngine          # the ctypes binding; here it loads the CPU library
    lib_path = os.path.join(os.path.dirname(os.path.abspath(oracle.__file__)), '_build', 'libremd_cpu.so')
    if not os.path.exists(lib_path):
        oracle.build()

    def make(md_steps):
        eng = HipEngine(lib_path=lib_path)
        eng.is_device = False
        sampler, _ = build_sampler(n_replicas, eng, None, md_steps)
        return sampler, eng

    s, eng = make(2)
    threads = int(eng.lib.remd_cpu_num_threads())
    t0 = time.perf_counter()
    s.run(1)                                                # builds lists, meshes, FFT tables; times 2 steps
    probe = time.perf_counter() - t0
    per_step = max(1e-4, float(s._timing_data['propagation_seconds']) / 2.0)
    eng.close()
    k = int(max(2, min(MD_STEPS, budget_s / per_step)))
    s, eng = make(k)
    s.run(1)                                                # warm-up iteration (iteration 0 energies, first lists)
    t0 = time.perf_counter()
    s.run(1)
    sample_s = time.perf_counter() - t0
    td = s._timing_data
    t_mix, t_prop, t_en = float(td['mixing_seconds']), float(td['propagation_seconds']), float(td['energy_seconds'])
    per_iter = t_mix + t_prop * (MD_STEPS / float(k)) + t_en
    eng.close()
    # SURVEY 8(d) asks for ALL host cores.  The port parallelises over replicas (one per thread: the shape of the reference's mpiplus
    # distribution), so the box is full when the ensemble has as many replicas as it has hardware threads: a second bounded sample
    # times such an ensemble and reports it in the metric's 24-replica units -- what these host cores deliver on this workload
    full = None
    if all_cores and threads > n_replicas:
        try:
            r_all = int(threads)
            eng2 = HipEngine(lib_path=lib_path)
            eng2.is_device = False
            k2 = max(2, k // 2)
            s2, _ = build_sampler(r_all, eng2, None, k2)
            s2.run(1)
            t0 = time.perf_counter()
            s2.run(1)
            sample2 = time.perf_counter() - t0
            td2 = s2._timing_data
            per_iter2 = float(td2['mixing_seconds']) + float(td2['propagation_seconds']) * (MD_STEPS / float(k2)) + float(td2['energy_seconds'])
            eng2.close()
            full = dict(kind='port-f64-all-threads', value=(r_all / float(n_replicas)) / per_iter2, unit='iterations/s in 24-replica units',
                        replicas=r_all, cores=r_all, seconds_per_iteration_of_that_ensemble=per_iter2,
                        sample='one iteration of a %d-replica ensemble (one replica per hardware thread) with %d of %d MD steps, %.1f s measured'
                               % (r_all, k2, MD_STEPS, sample2))
        except Exception as exc:
            full = dict(error='%s: %s' % (type(exc).__name__, exc))
    return dict(value=1.0 / per_iter, unit='iterations/s', cores=min(threads, n_replicas), kind='port', all_host_threads=full,
                threads_available=threads, seconds_per_iteration=per_iter,
                sample='one full mix -> propagate -> u_kl iteration of the %d-replica AlanineDipeptideExplicit ensemble on '
                       'libremd_cpu.so (same C ABI, f64, OpenMP over replicas) with %d of %d MD steps (%.1f s measured; '
                       'propagation scaled x%.1f, mixing %.4f s and energy matrix %.3f s in full; probe %.1f s).  %d of the '
                       'box\'s %d hardware threads are busy: one per replica, the shape of the reference\'s mpiplus distribution; '
                       'the rest idle (a scalar f64 port -- a lower bound on what OpenMM\'s vectorised single-precision CPU '
                       'platform would do)' %
                       (n_replicas, k, MD_STEPS, sample_s, MD_STEPS / float(k), t_mix, t_en, probe, min(threads, n_replicas), threads))


def build_dhfr_sams(n_replicas, engine, comm, md_steps):
    """BASELINE config 5: testsystems.DHFRExplicit (23 558 atoms), 128 states on a temperature ladder, SAMSSampler with the global
    jump (sams.py:477-501), g-BAOAB 2 fs."""
    from openmmtools_amd import testsystems, states, mcmc, unit
    from openmmtools_amd.multistate import SAMSSampler
    dh = testsystems.DHFRExplicit()
    ths = [states.ThermodynamicState(dh.system, t) for t in np.geomspace(300.0, 400.0, n_replicas)]
    move = mcmc.LangevinSplittingDynamicsMove(timestep=2.0 * unit.femtosecond, collision_rate=1.0 / unit.picosecond, n_steps=md_steps,
                                              reassign_velocities=True, splitting='V R R O R R V')
    s = SAMSSampler(mcmc_moves=move, number_of_iterations=10 ** 9, engine=engine, seed=SEED, comm=comm)
    s.create(ths, [states.SamplerState(dh.positions, box_vectors=dh.system.getDefaultPeriodicBoxVectors())] * n_replicas)
    return s


def run_shapes(args, world, rank, local_rank, comm, sync, stream, budget_s=240.0, out=None):
    """The other ensemble shapes on the same ranks (every rank calls this: the samplers hold collectives).  Returns the dict for the
    JSON line on rank 0.  A shape that fails is reported with its error instead of taking the main line down."""
    import torch
    from openmmtools_amd._engine import HipEngine
    # rank 0 arrives here tens of seconds after the others (roof microbenchmark, CPU baseline): the clock that decides what is skipped
    # starts behind a barrier, and the decision itself is collective (a rank that skips a shape another rank starts leaves
    # that shape's collectives unmatched)
    sync()
    t_start = time.perf_counter()
    out = {} if out is None else out       # (caller-owned: the watchdog of main() reports what finished)

    def elapsed_on_the_slowest_rank():
        e = time.perf_counter() - t_start
        if world > 1:
            import torch.distributed as dist
            t = torch.tensor([e], dtype=torch.float64, device='cuda' if dist.get_backend() == 'nccl' else 'cpu')
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            e = float(t.item())
        return e
    specs = [('strong24_alanine', 24, 'alanine', 3, 1), ('strong128_alanine', 128, 'alanine', 3, 1),
             ('config5_dhfr128_sams', 128, 'dhfr', 1 if world == 1 else 2, 1)]
    for name, R, kind, n_it, n_warm in specs:
        if R < world or (name == 'strong24_alanine' and world == 1):
            continue
        if elapsed_on_the_slowest_rank() > budget_s:
            out[name] = dict(skipped='time budget of the extra shapes used up')
            continue
        try:
            engine = HipEngine(device=local_rank, stream=stream)
            sampler = build_sampler(R, engine, comm, args.md_steps)[0] if kind == 'alanine' else build_dhfr_sams(R, engine, comm, args.md_steps)
            sampler.run(n_warm)
            sync()
            t0 = time.perf_counter()
            sampler.run(n_it)
            sync()
            elapsed = time.perf_counter() - t0
            td = sampler._timing_data
            parts = [float(td.get(k, 0.0)) for k in ('mixing_seconds', 'propagation_seconds', 'energy_seconds', 'allgather_seconds')]
            if world > 1:
                import torch.distributed as dist
                dev = 'cuda' if dist.get_backend() == 'nccl' else 'cpu'
                t = torch.tensor([elapsed] + parts, dtype=torch.float64, device=dev)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                elapsed, parts = float(t[0].item()), [float(v) for v in t[1:].tolist()]
            it_per_s = n_it / elapsed
            out[name] = dict(value=it_per_s * (R / float(REPLICAS_PER_GPU)) if kind == 'alanine' else it_per_s,
                             unit='iterations/s in 24-replica units (R / 24 x ensemble iterations/s)' if kind == 'alanine'
                                  else 'iterations/s of the 128-replica ensemble',
                             scaling='strong', replicas_total=R, replicas_on_rank0=int(sampler._r_count), n_gpus=world, world_size_seen=world,
                             backend=(os.environ.get('REMD_BENCH_BACKEND', 'nccl') if world > 1 else 'none'),
                             iterations=n_it, warmup=n_warm, md_steps=args.md_steps, ms_per_iteration=1e3 * elapsed / n_it,
                             slowest_rank_last_iteration_ms=dict(mix=1e3 * parts[0], propagate=1e3 * parts[1], ukl=1e3 * parts[2],
                                                                 allgather=1e3 * parts[3]),
                             mix=('swap-all, R^3 attempts, replicated on every rank' if kind == 'alanine' else 'SAMS global jump, R x K'))
            engine.close()
        except Exception as exc:                       # never fail the main line for an extra shape
            out[name] = dict(error='%s: %s' % (type(exc).__name__, exc))
    return out


def finish_line(out, rank, shapes_fn, limit_s, write=None, end_process=None, align=None):
    """Print the ONE JSON line (rank 0), after the extra ensemble shapes if there are any.  The shapes run LAST and under a watchdog: the
    headline part of `out` is complete when this is called, and an extra shape that stalls (a device-side poll that runs out takes seconds
    per MD step) must not keep it from being printed.  The watchdog thread runs while the main thread sits in a C call (ctypes releases
    the GIL); it prints the line with the shapes that did finish and ends the process -- on every rank, each by its own timer.
    shapes_fn(dict) fills the dict shape by shape (every rank calls it: the samplers hold collectives); None: no extra shapes.
    align: called on every rank right before the timer starts (a barrier), so that no rank's watchdog ends its process while another
    rank, whose timer started later, is still inside a collective of a shape."""
    import threading
    write = write or (lambda text: print(text, flush=True))
    end_process = end_process or (lambda: os._exit(0))
    printed = threading.Lock()

    def emit(note=None):
        if not printed.acquire(blocking=False):
            return False
        if rank == 0:
            if note and out.get('shapes') is not None:
                out['shapes']['watchdog'] = note
            write(json.dumps(out, default=float))
        return True

    if shapes_fn is not None:
        shapes = {}
        if rank == 0:
            out['shapes'] = shapes

        def expired():
            if emit('the extra shapes did not finish within %.0f s: the line is printed without the rest and the process ends' % limit_s):
                end_process()
        if align is not None:
            align()                                    # (a barrier: every rank's timer then runs from the same moment)
        dog = threading.Timer(limit_s, expired)
        dog.daemon = True
        dog.start()
        try:
            shapes_fn(shapes)
        except Exception as exc:                       # (whatever happens in the extras, the line is printed)
            shapes['error'] = '%s: %s' % (type(exc).__name__, exc)
        dog.cancel()
    emit()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--md-steps', type=int, default=MD_STEPS, help='MD steps per iteration (500 = BASELINE)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-shapes', action='store_true', help='skip the extra ensemble shapes (strong-24 / strong-128 / config 5)')
    ap.add_argument('--replicas-total', type=int, default=0,
                    help='strong scaling: one ensemble of this many replicas sharded over the ranks (24 = the BASELINE '
                         'metric as written, 128 = the north_star efficiency target); 0 = weak scaling, 24 per GPU')
    args = ap.parse_args()

    import torch
    rank = int(os.environ.get('RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit('launch with: python -m torch.distributed.run --nnodes=1 --nproc-per-node %d '
                             '--master-addr 127.0.0.1 --master-port 29511 bench.py --gpus %d ...' % (args.gpus, args.gpus))
    if os.environ.get('REMD_BENCH_SHARE_GPU'):       # functional test of the N > 1 flow on a one-GPU box (with gloo)
        local_rank = 0
    torch.cuda.set_device(local_rank)
    comm = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        backend = os.environ.get('REMD_BENCH_BACKEND', 'nccl')     # "nccl" is RCCL on ROCm
        if backend == 'nccl':
            dist.init_process_group('nccl', rank=rank, world_size=world, device_id=torch.device('cuda', local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
        from openmmtools_amd.multistate.comm import TorchDistributedComm
        comm = TorchDistributedComm()

    from openmmtools_amd._engine import HipEngine
    stream = torch.cuda.current_stream().cuda_stream
    engine = HipEngine(device=local_rank, stream=stream)
    strong = args.replicas_total > 0
    n_replicas = args.replicas_total if strong else REPLICAS_PER_GPU * world
    if n_replicas < world:
        raise SystemExit('--replicas-total must be at least the number of ranks')
    sampler, ts = build_sampler(n_replicas, engine, comm, args.md_steps)
    n_local = sampler._r_count                  # replicas on this rank (block partition, comm.py)
    # the Ewald split the engine runs (HipEngine.ewald_split -> system.system_to_desc): Coulomb range, alpha, mesh
    from openmmtools_amd.system import system_to_desc
    _d = system_to_desc(ts.system, ewald_split=engine.ewald_split)
    ewald = dict(split=str(engine.ewald_split), lj_cutoff_nm=float(_d['cutoff']), coulomb_cutoff_nm=float(_d.get('coulomb_cutoff', _d['cutoff'])),
                 alpha_per_nm=float(_d['ewald_alpha']), pme_grid=[int(g) for g in _d['pme_grid']], ewald_error_tolerance=1e-5,
                 note="'auto': the direct-space erfc sum runs beyond the 1.0 nm Lennard-Jones cutoff and the mesh shrinks by the same "
                      "tolerance rule (reference split: alpha 3.289 / nm, 75 x 75 x 72); u_kl parity against OpenMM's own numbers at this "
                      "split: tests/test_openmm_fixture.py")
    n_atoms = ts.system.getNumParticles()

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
            torch.cuda.synchronize()

    sampler.run(args.warmup)
    engine.profile_reset()
    engine.profile_enable(True, 'nonbonded|pme_xy')   # asynchronous HIP events around the two heaviest kernel classes only
    sync()
    t0 = time.perf_counter()
    sampler.run(args.steps)
    sync()
    elapsed = time.perf_counter() - t0
    engine.profile_enable(False)
    timing_of_timed_region = dict(sampler._timing_data)       # (the untimed integrator-profiling iteration below would overwrite it)
    if world > 1:
        import torch.distributed as dist
        t = torch.tensor([elapsed], dtype=torch.float64, device='cuda' if dist.get_backend() == 'nccl' else 'cpu')
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    out = None
    if rank == 0:
        it_per_s = args.steps / elapsed
        value = it_per_s * (n_replicas / REPLICAS_PER_GPU)
        # one force evaluation = the Coulomb launch (class 'nonbonded') + the LJ sub-system launch ('nonbonded_lj')
        n_launch, ms = engine.profile_get('nonbonded')
        n_lj, ms_lj = engine.profile_get('nonbonded_lj')
        n_xy, ms_xy = engine.profile_get('pme_xy')
        roof_nb = roof_xy = None
        if n_launch > 0:
            phases = engine.phases_active() if hasattr(engine, 'phases_active') else 1
            flops = FLOP_PER_ATOM_NONBONDED * n_atoms * n_local / float(phases)      # a launch covers one phase's share of the replicas
            avg_ms = (ms + ms_lj) / n_launch
            achieved = flops / (avg_ms * 1e-3) / 1e12
            roof_nb = dict(kernel='nonbonded_sci2_kernel', bound='mfma', achieved=achieved, peak=FP32_PEAK_TFLOPS, unit='TFLOP/s',
                           frac=achieved / FP32_PEAK_TFLOPS, traffic=pmc_traffic_bytes('nonbonded_sci2_kernel'),
                           launches=n_launch, avg_launch_ms=avg_ms, total_ms=ms + ms_lj, phases=phases, replicas_per_launch=n_local / float(phases),
                           traffic_source=pmc_traffic_source(), traffic_source_is_this_kernel=pmc_traffic_is_current(),
                           note='fp32 VALU kernel (no MFMA: "mfma" is the contract\'s name for the compute roof); peak = FP32 vector rate = '
                                'f32-input MFMA rate (157.3 TFLOP/s); algorithmic work = 10 kflop/atom (SURVEY 8(d): ~210 pairs per atom '
                                'inside the reference\'s 1.0 nm cutoff x ~48 flop -- the rebalanced Ewald split evaluates 1.43x as many '
                                'Coulomb pairs, from a force table, which is NOT counted); one launch evaluates the Coulomb system and the '
                                'LJ sub-system (every pair once: Newton\'s third law on per-tile union lists); the timed scope also holds '
                                'the sorted-slot force scatter.  The duration is the one next to the mesh kernels of the other stream '
                                '(stand-alone: 0.076 ms); traffic is an offline PMC figure, see traffic_source')
        if n_xy > 0:
            # algorithmic bytes (SURVEY 8(d) "grid traffic 8 B x G per pass"): the half spectrum [nz/2+1][nx][ny] complex f32 of
            # every replica is read once and written once by the plane-resident XY pass
            nx, ny, nz = ewald['pme_grid']
            nbytes = 2.0 * 8.0 * (nz // 2 + 1) * nx * ny * n_local / float(engine.phases_active() if hasattr(engine, 'phases_active') else 1)
            avg_ms = ms_xy / n_xy
            achieved = nbytes / (avg_ms * 1e-3) / 1e9
            xy_name = 'pme_xy_pow2_kernel' if (nx == ny and nx in (64, 128) and os.environ.get('REMD_PME_POW2', '3') not in ('0', '2')) else 'pme_xy_fused_kernel'
            roof_xy = dict(kernel=xy_name, bound='hbm', achieved=achieved, peak=HBM_PEAK_GBS, unit='GB/s',
                           frac=achieved / HBM_PEAK_GBS, traffic=pmc_traffic_bytes(xy_name), launches=n_xy,
                           avg_launch_ms=avg_ms, total_ms=ms_xy,
                           note='forward x, forward y, influence function, inverse y, inverse x on an LDS-resident plane: one read + '
                                'one write of the half spectrum (round 6: radix-8 x 8 butterflies in registers on 64 x 64 planes, '
                                'pme_pow2.h; other sizes: scheduled mixed-radix stages); it shares the chip with the pair kernel on '
                                'the other stream')
        # the integrator chain against its own roof (fused bound 64 B/atom: read x, v, f, 1/m, write x, v)
        # (timed in one extra, untimed iteration: events around the chain launches sit on the critical path of a step and
        # would slow the timed region by ~6 %)
        n_ch, ms_ch = 0, 0.0
        if world == 1:                                   # (an extra iteration is a collective under N > 1: single rank only)
            engine.profile_reset()
            engine.profile_enable(True, 'integrate_chain')
            sampler.run(1)
            torch.cuda.synchronize()
            engine.profile_enable(False)
            n_ch, ms_ch = engine.profile_get('integrate_chain')
            n_own, ms_own = engine.profile_get('integrate_chain_own')
        roof_ch = None
        if n_ch > 0:
            # launch-to-end (HIP events) holds the prologue's wait for the direct-space stream's "forces complete" flag; the roofline
            # number uses the launch's OWN time: flag seen -> end, from wall-clock stamps inside the kernel (workgroup (0, 0))
            avg_launch_ms = ms_ch / n_ch
            avg_ms = ms_own / n_own if n_own > 0 else avg_launch_ms
            achieved = 64.0 * n_atoms * n_local / float(engine.phases_active() if hasattr(engine, 'phases_active') else 1) / (avg_ms * 1e-3) / 1e9
            roof_ch = dict(kernel='integrate_chain_kernel', bound='hbm', achieved=achieved, peak=HBM_PEAK_GBS, unit='GB/s',
                           frac=achieved / HBM_PEAK_GBS, traffic=pmc_traffic_bytes('integrate_chain_kernel'), launches=n_ch,
                           avg_launch_ms=avg_ms, avg_launch_to_end_ms=avg_launch_ms, total_ms=ms_ch,
                           note='one launch per MD step: V | sum(m v) + barrier over the replica\'s workgroups | C V R R O R R | PME '
                                'binning.  avg_launch_ms = the kernel\'s own time from "forces complete" seen to its end (stamps inside '
                                'the kernel); avg_launch_to_end_ms = what HIP events / rocprofv3 report, which also holds the wait '
                                'for the direct-space stream in the prologue.  One thread per rigid water / X-H cluster / free '
                                'atom with x, v in registers: 216 workgroups, bound by the latency of the dependent SETTLE / '
                                'RATTLE arithmetic and the barrier, not by HBM')
        # the contract asks for the dominant kernel: the class with the larger accumulated time in the timed region
        cands = [r for r in (roof_nb, roof_xy) if r]
        cands.sort(key=lambda r: -r['total_ms'])
        roof = cands[0] if cands else None
        roof2 = cands[1] if len(cands) > 1 else None
        out = dict(metric='REMD iterations/s (propagate+u_kl+mix), 24-replica AlanineDipeptideExplicit' +
                          (' per GPU' if not strong else ' units, one %d-replica ensemble over all GPUs' % n_replicas),
                   value=value, unit='iterations/s', n_gpus=world, steps=args.steps, warmup=args.warmup,
                   ms_per_step=1e3 * elapsed / args.steps, higher_is_better=True, scaling='strong' if strong else 'weak',
                   vs_baseline=None,
                   dtype='f32', data='synthetic',
                   config=dict(workload='testsystems.AlanineDipeptideExplicit (2269 atoms, PME) parallel tempering, '
                                        'logspace(300K,600K), g-BAOAB V R R O R R V 2 fs, %d MD steps/iteration, swap-all'
                                        % args.md_steps,
                               replicas_per_gpu=(n_replicas / float(world)), replicas_total=n_replicas, md_steps=args.md_steps,
                               mode=('strong: one %d-replica ensemble' % n_replicas) if strong else 'weak: 24 replicas per GPU',
                               parallelism='replica-sharded x%d' % world, seed=SEED, ewald=ewald,
                               phases_per_gpu=(engine.phases_active() if hasattr(engine, 'phases_active') else 1)),
                   timing=timing_of_timed_region, roofline=roof, roofline_secondary=roof2, roofline_integrator=roof_ch, shapes=None)
        try:
            # achievable roofs of THIS box (STREAM triad past the Infinity Cache, FMA chains), SURVEY 8(d)
            out['measured_roofs'] = engine.roof_microbench()
            if roof and roof['unit'] == 'TFLOP/s':
                roof['frac_of_measured_pk_fma'] = roof['achieved'] / out['measured_roofs']['pk_fma_f32_tflop_per_s']
            out['measured_roofs']['note'] = ('FMA chains at full occupancy; below the 157.3 TFLOP/s spec because the chip does not hold 2.4 GHz under '
                                             'this load: shader_clock_ghz_under_fma_load is measured inside the microbenchmark (cycle counter against '
                                             'the 100 MHz wall clock) and fma_f32_peak_at_that_clock = 157.3 x clock / 2.4; packed FMA issues at half '
                                             'the rate of plain FMA, so it gives no extra throughput on gfx950')
            for r_ in (roof, roof2, roof_ch):
                if r_ and r_['unit'] == 'GB/s':
                    r_['frac_of_measured_stream'] = r_['achieved'] / out['measured_roofs']['stream_triad_gb_per_s']
        except Exception as exc:                      # a measurement extra: never fail the bench line for it
            out['measured_roofs'] = dict(error=str(exc))
        if not args.no_cpu_baseline and world == 1:
            engine.close()
            out['cpu_baseline'] = cpu_baseline()
        else:
            out['cpu_baseline'] = None

    shapes_fn = None
    if not args.no_shapes and args.replicas_total == 0:
        def shapes_fn(shapes):
            run_shapes(args, world, rank, local_rank, comm, sync, stream, out=shapes)
    finish_line(out, rank, shapes_fn, float(os.environ.get('REMD_BENCH_SHAPES_LIMIT_S', '420')), align=sync if world > 1 else None)
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()

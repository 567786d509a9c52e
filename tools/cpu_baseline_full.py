"""The CPU baseline over the FULL 500 MD steps of an iteration, once per round (bench.py's line keeps its bounded
sample): the same 24-replica mix -> propagate -> u_kl iteration on oracle/_build/libremd_cpu.so (reference split of the Ewald sum).
usage: python tools/cpu_baseline_full.py > profiles/rNN_cpu_baseline_full.json"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import oracle
from openmmtools_amd._engine import HipEngine

lib_path = os.path.join(os.path.dirname(os.path.abspath(oracle.__file__)), '_build', 'libremd_cpu.so')
if not os.path.exists(lib_path):
    oracle.build()
eng = HipEngine(lib_path=lib_path)
eng.is_device = False
sampler, _ = bench.build_sampler(bench.REPLICAS_PER_GPU, eng, None, bench.MD_STEPS)
threads = int(eng.lib.remd_cpu_num_threads())
sampler.run(1)                       # iteration 0 energies, lists, meshes
t0 = time.perf_counter()
sampler.run(1)
wall = time.perf_counter() - t0
td = sampler._timing_data
print(json.dumps(dict(kind='port', library='libremd_cpu.so (f64, OpenMP over replicas)', ewald_split=str(eng.ewald_split), md_steps=bench.MD_STEPS,
                      replicas=bench.REPLICAS_PER_GPU, threads_used=min(threads, bench.REPLICAS_PER_GPU), threads_available=threads,
                      seconds_per_iteration=wall, iterations_per_s=1.0 / wall, mixing_seconds=float(td['mixing_seconds']),
                      propagation_seconds=float(td['propagation_seconds']), energy_seconds=float(td['energy_seconds']))))
eng.close()

"""Does the chip run two half-ensembles faster than one whole one?  The headline step has ~70 us of dependent-launch latency that does not
scale with replicas (profiles/r04_scaling_projection.md); replicas are independent between mixes, so G handles of R/G replicas each,
propagated concurrently from G host threads (ctypes releases the GIL), can fill each other's latency bubbles.
usage: python tools/group_overlap.py [R=24] [system=alanine] [groups ...=1 2 3]"""
import os, sys, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from openmmtools_amd import testsystems as ts
from openmmtools_amd.system import system_to_desc
from openmmtools_amd._engine import HipEngine
KB = 0.008314462618153242
R = int(sys.argv[1]) if len(sys.argv) > 1 else 24
name = sys.argv[2] if len(sys.argv) > 2 else 'alanine'
groups = [int(a) for a in sys.argv[3:]] or [1, 2, 3]
al = {'alanine': ts.AlanineDipeptideExplicit, 'hostguest': ts.HostGuestExplicit, 'dhfr': ts.DHFRExplicit}[name]()
box = np.diag(al.system.getDefaultPeriodicBoxVectors())
d = system_to_desc(al.system, ewald_split='auto')
beta = 1 / (KB * np.geomspace(300.0, 600.0, R))
rng = np.random.default_rng(7)
x0 = np.tile(al.positions, (R, 1, 1)) + rng.normal(0, 0.002, (R,) + al.positions.shape)
n_steps = int(os.environ.get('GO_STEPS', '500'))
ref = None
for G in groups:
    cuts = np.linspace(0, R, G + 1).round().astype(int)
    engs = []
    for g in range(G):
        a, b = cuts[g], cuts[g + 1]
        e = HipEngine(lib_path=os.environ.get("AB_LIB") or None, ewald_split="auto")
        e.set_system(d); e.set_states(beta)
        e.set_integrator('V R R O R R V', 0.002, 1.0, n_steps, True, 1e-8)
        e.seed(11)
        e.set_replicas(R, int(a), x0[a:b], None, np.tile(box, (b - a, 1)), np.arange(R))
        engs.append(e)

    def run(it):
        th = [threading.Thread(target=e.propagate, args=(it,)) for e in engs]
        t = time.perf_counter()
        for q in th: q.start()
        for q in th: q.join()
        return 1e3 * (time.perf_counter() - t)
    run(0)
    ms = [run(it) for it in range(1, 6)]
    own = [e.last_timing()['propagate_ms'] for e in engs]
    x = np.concatenate([e.get_replicas()[0] for e in engs])
    if ref is None:
        ref = x
    print('%s R %d  groups %d %s: wall ms per %d steps  min %.2f  all %s | per-handle device ms %s | positions identical to first: %s' % (
        name, R, G, np.diff(cuts).tolist(), n_steps, min(ms), ' '.join('%.1f' % m for m in ms), ' '.join('%.1f' % o for o in own),
        bool(np.array_equal(x, ref))), flush=True)
    for e in engs: e.close()

"""Round-6 probe: G handles of R/G replicas, driven sequentially or from G host threads; prints wall ms per propagate and a digest of
the final positions per GLOBAL replica so that runs in different processes (different environment switches) can be compared.
usage: python tools/phase_probe.py R G mode[seq|thr|many] [system] ; env GO_STEPS, GO_ITERS"""
import os, sys, time, threading, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from openmmtools_amd import testsystems as ts
from openmmtools_amd.system import system_to_desc
from openmmtools_amd._engine import HipEngine
KB = 0.008314462618153242
_extra = []
if os.environ.get('GO_EXTRA_STREAMS'):          # other raised-priority streams in the process before the engines' (as torch's NCCL stream)
    import torch
    _extra = [torch.cuda.Stream(priority=-1) for _ in range(int(os.environ['GO_EXTRA_STREAMS']))]
    for q in _extra:
        with torch.cuda.stream(q): torch.zeros(8, device='cuda').sum().item()
R = int(sys.argv[1]); G = int(sys.argv[2]); mode = sys.argv[3]
name = sys.argv[4] if len(sys.argv) > 4 else 'alanine'
al = {'alanine': ts.AlanineDipeptideExplicit, 'hostguest': ts.HostGuestExplicit, 'dhfr': ts.DHFRExplicit}[name]()
box = np.diag(al.system.getDefaultPeriodicBoxVectors())
d = system_to_desc(al.system, ewald_split='auto')
beta = 1 / (KB * np.geomspace(300.0, 600.0, R))
rng = np.random.default_rng(7)
x0 = np.tile(al.positions, (R, 1, 1)) + rng.normal(0, 0.002, (R,) + al.positions.shape)
n_steps = int(os.environ.get('GO_STEPS', '500')); iters = int(os.environ.get('GO_ITERS', '4'))
cuts = np.linspace(0, R, G + 1).round().astype(int)
engs = []
for g in range(G):
    a, b = cuts[g], cuts[g + 1]
    e = HipEngine(ewald_split="auto", lib_path=os.environ.get("AB_LIB") or None)
    e.set_system(d); e.set_states(beta)
    e.set_integrator('V R R O R R V', 0.002, 1.0, n_steps, True, 1e-8)
    e.seed(11)
    e.set_replicas(R, int(a), x0[a:b], None, np.tile(box, (b - a, 1)), np.arange(R))
    engs.append(e)
if os.environ.get('GO_PHASES'):
    for e in engs: e.set_phases(int(os.environ['GO_PHASES']))

def run(it):
    t = time.perf_counter()
    if mode == 'many':
        HipEngine.propagate_many(engs, it)
    elif mode == 'thr':
        th = [threading.Thread(target=e.propagate, args=(it,)) for e in engs]
        for q in th: q.start()
        for q in th: q.join()
    else:
        for e in engs: e.propagate(it)
    return 1e3 * (time.perf_counter() - t)
ms = [run(it) for it in range(iters)]
own = [e.last_timing()['propagate_ms'] for e in engs]
x = np.concatenate([e.get_replicas()[0] for e in engs])
dig = [hashlib.sha1(np.ascontiguousarray(x[r]).tobytes()).hexdigest()[:8] for r in range(R)]
print('%s R %d G %d %s steps %d env[%s]: ms %s | device ms %s | digest all %s first %s last %s' % (
    name, R, G, mode, n_steps, ' '.join('%s=%s' % (k, v) for k, v in sorted(os.environ.items()) if k.startswith(('REMD_', 'GPU_MAX', 'GO_PHASES', 'GO_EXTRA'))),
    ' '.join('%.1f' % m for m in ms), ' '.join('%.1f' % o for o in own),
    hashlib.sha1(x.tobytes()).hexdigest()[:10], dig[0], dig[-1]), flush=True)
print('  per-replica', ' '.join(dig), flush=True)
for e in reversed(engs): e.close()

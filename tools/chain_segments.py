"""Where the integrator chain's own time goes: wall-clock stamps of workgroup (0, 0) after every token of the launch's program
(a library built with -DCHAIN_STAMPS: tools/build_variant.sh stamps integrate.hip -DCHAIN_STAMPS).
usage: AB_LIB=$PWD/openmmtools_amd/libremd_hip_stamps.so python tools/chain_segments.py [R] [splitting]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from openmmtools_amd import testsystems as ts
from openmmtools_amd.system import system_to_desc
from openmmtools_amd._engine import HipEngine
KB = 0.008314462618153242
R = int(sys.argv[1]) if len(sys.argv) > 1 else 24
splitting = sys.argv[2] if len(sys.argv) > 2 else 'V R R O R R V'
al = ts.DHFRExplicit() if os.environ.get('SEG_SYSTEM') == 'dhfr' else ts.AlanineDipeptideExplicit()
box = np.diag(al.system.getDefaultPeriodicBoxVectors())
eng = HipEngine(lib_path=os.environ.get('AB_LIB') or None)
eng.set_system(system_to_desc(al.system, ewald_split='auto')); eng.set_states(1 / (KB * np.linspace(300.0, 600.0, R)))
eng.set_integrator(splitting, 0.002, 5.0, int(os.environ.get('SEG_STEPS', '500')), True, 1e-8)
eng.seed(1)
eng.set_replicas(R, 0, np.tile(al.positions, (R, 1, 1)), None, np.tile(box, (R, 1)), np.arange(R))
eng.propagate(0)
eng.profile_enable(True, 'integrate_chain'); eng.profile_reset()      # (level 2 would split the chain at the momentum sum)
eng.propagate(1)
n, own = eng.profile_get('integrate_chain_own')
print('launches', n, 'own us per launch', round(1e3 * own / max(n, 1), 2))
names = ['prologue'] + ['tok%d' % t for t in range(34)] + ['epilogue']
# round 5: finer stamps in the unused token slots (each drains the wavefront's memory operations first, so they over-count a little)
names[1 + 20] = 'tok20 (unit table, temperature arrived)'
names[1 + 21] = 'tok21 (positions, velocities, masses arrived)'
names[1 + 22] = 'tok22 (forces arrived + first kick)'
for k in range(3):
    names[1 + 24 + k] = 'tok%d (stores of atom %d drained)' % (24 + k, k)
names[1 + 27] = 'tok27 (mesh-column bins by the workgroup)'
for k, name in enumerate(names):
    _, ms = eng.profile_get('integrate_chain_seg%d' % k)
    if ms > 0:
        print('%-48s %7.2f us' % (name, 1e3 * ms / max(n, 1)))

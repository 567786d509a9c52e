"""Diagnostic (GPU box): per-component energies and forces of AlanineDipeptideExplicit, device vs oracle."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, math
from openmmtools_amd import testsystems as ts
from openmmtools_amd.system import system_to_desc
from openmmtools_amd._engine import HipEngine
from oracle.forcefield import ForceFieldOracle
from oracle.md_oracle import ONE_4PI_EPS0
KB = 0.008314462618153242
al = ts.AlanineDipeptideExplicit()
desc = system_to_desc(al.system)
eng = HipEngine()
eng.set_system(desc); eng.set_states(np.array([1/(KB*300.0)])); eng.set_integrator('V R O R V', 0.001, 1.0, 1, True, 1e-8)
box = np.diag(al.system.getDefaultPeriodicBoxVectors())
eng.set_replicas(1, 0, al.positions[None], None, box[None], np.zeros(1, int))
comp = eng.energy_components()[0]
ff = ForceFieldOracle(desc)
x = torch.tensor(eng.get_replicas()[0][0]); bt = torch.tensor(box)
pairs = ff._pairs(x.numpy(), box)
ref = {}
ref['bonded_total'] = float(ff._bonded(x))
ref['nonbonded_direct'] = float(ff._pair_terms(x, bt, pairs, 1.0, 1.0))
i, j = ff.exc_atoms[:, 0], ff.exc_atoms[:, 1]
r = ff._min_image(x[j] - x[i], bt).norm(dim=1); p = torch.tensor(ff.exc_params)
nz = (p[:, 0] != 0) | (p[:, 2] != 0)
sr6 = torch.where(nz, (p[:, 1] / r) ** 6, torch.zeros_like(r))
ref['exceptions'] = float((torch.where(nz, ONE_4PI_EPS0 * p[:, 0] / r, torch.zeros_like(r)) + 4 * p[:, 2] * sr6 * (sr6 - 1)).sum())
ref['ewald_exclusions'] = float(-(ONE_4PI_EPS0 * ff.q[i] * ff.q[j] * torch.erf(ff.alpha * r) / r).sum())
ref['pme_reciprocal'] = float(ff.pme_reciprocal(x, bt, ff.q))
V = box.prod()
ref['constants'] = ff.disp_coeff / V - ONE_4PI_EPS0 * ff.alpha / math.sqrt(math.pi) * float((ff.q ** 2).sum()) - ONE_4PI_EPS0 * math.pi * float(ff.q.sum()) ** 2 / (2 * ff.alpha ** 2 * V)
print('device', comp)
print('oracle', ref, 'dev bonded', comp['bonds'] + comp['angles'] + comp['torsions'])
e_ref, f_ref = ff.energy_forces(x.numpy(), box)
f = eng.get_forces()[0]
print('total', sum(comp.values()), e_ref)
err = np.abs(f - f_ref)
print('force max err', err.max(), 'at atom', np.unravel_index(err.argmax(), err.shape), 'rmse', np.sqrt((err**2).sum(1).mean()))
print('solute err', err[:22].max(), 'water err', err[22:].max())
worst = np.argsort(-err.max(1))[:10]; print('worst atoms', worst, err.max(1)[worst])

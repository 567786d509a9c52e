"""TEST INFRASTRUCTURE (oracle): f64 restatement of the GBSA (OBC2 + ACE surface term) energy as the reference's alchemical factory writes it
for OpenMM -- /root/reference/openmmtools/alchemy/alchemy.py:2144-2225 (_alchemically_modify_GBSAOBCForce): computed values I (pair sum)
and B (Born radius), a self term, a surface term and the pair term, every one with the factor lambda_electrostatics on the alchemical
particles.  At lambda = 1 (or without alchemical particles) it is OpenMM's GBSAOBCForce, which the factory replaces (:2144-2170).
Pinned by tests/test_gbsa.py against tests/golden/reference_gbsa.json: the reference's own expression strings evaluated by an interpreter
of the CustomGBForce semantics (tests/golden/make_golden_gbsa.py).  Forces by autograd.  NoCutoff only.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
import numpy as np
import torch

K_E = 138.935485              # the factory's literal (alchemy.py:2205, 2209), not constants.ONE_4PI_EPS0
OFFSET = 0.009                # :2192
SASA = 28.3919551             # :2207: 4 pi x 2.25936 kJ/mol/nm^2


def gbsa_energy_torch(x, charge, radius, scale, alchemical, lam, solute_dielectric=1.0, solvent_dielectric=78.5, sasa=True, return_parts=False):
    n = x.shape[0]
    q = torch.as_tensor(charge, dtype=torch.float64); R = torch.as_tensor(radius, dtype=torch.float64)
    sc = torch.as_tensor(scale, dtype=torch.float64); a = torch.as_tensor(alchemical, dtype=torch.float64)
    s = lam * a + (1.0 - a)                                            # (lambda_electrostatics*alchemical + (1-alchemical))
    orr = R - OFFSET
    sr = sc * orr
    eye = torch.eye(n, dtype=torch.bool)
    d = x[:, None, :] - x[None, :, :]
    r = torch.sqrt((d * d).sum(-1) + eye.double())                     # (diagonal: 1, masked below)
    or1, sr2 = orr[:, None], sr[None, :]
    U = r + sr2
    D = torch.abs(r - sr2)
    L = torch.maximum(or1.expand(n, n), D)
    C = 2.0 * (1.0 / or1 - 1.0 / L) * (sr2 - r - or1 >= 0).double()
    H = (r + sr2 - or1 >= 0).double() * 0.5 * (1.0 / L - 1.0 / U + 0.25 * (r - sr2 ** 2 / r) * (1.0 / U ** 2 - 1.0 / L ** 2) + 0.5 * torch.log(L / U) / r + C)
    I = (s[None, :] * H).masked_fill(eye, 0.0).sum(1)
    psi = I * orr
    B = 1.0 / (1.0 / orr - torch.tanh(psi - 0.8 * psi ** 2 + 4.85 * psi ** 3) / R)
    tau = 1.0 / solute_dielectric - 1.0 / solvent_dielectric
    e = (-0.5 * K_E * tau * s * q ** 2 / B).sum()
    if sasa:
        e = e + (s * SASA * (R + 0.14) ** 2 * (R / B) ** 6).sum()
    BB = B[:, None] * B[None, :]
    f = torch.sqrt(r ** 2 + BB * torch.exp(-r ** 2 / (4.0 * BB)))
    pair = (-K_E * tau * (s * q)[:, None] * (s * q)[None, :] / f).masked_fill(eye, 0.0)
    e = e + 0.5 * pair.sum()
    return (e, I, B) if return_parts else e


def gbsa_energy_forces(x, charge, radius, scale, alchemical, lam, solute_dielectric=1.0, solvent_dielectric=78.5, sasa=True, forces=True):
    xt = torch.tensor(np.asarray(x, dtype=np.float64), requires_grad=forces)
    e = gbsa_energy_torch(xt, charge, radius, scale, alchemical, lam, solute_dielectric, solvent_dielectric, sasa)
    if not forces:
        return float(e.detach()), None
    (g,) = torch.autograd.grad(e, xt)
    return float(e.detach()), -g.numpy()

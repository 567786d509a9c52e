"""The matrix-core XY pass of the PME mesh (csrc/dft_mfma.hip): 2-D DFTs of n x n <= 80 x 80 planes as split-f16 MFMA
products.  Checked against numpy's f64 FFT (the bar is the accuracy of an f32 FFT: 1e-6 of the largest element), on
planes of very different magnitude in one launch (the power-of-two scaling is per plane), and end to end against the FFT
kernels it replaces (REMD_PME_XY_MFMA=0) on AlanineDipeptideExplicit's 75 x 75 x 72 mesh."""
import numpy as np
import pytest
from openmmtools_amd import testsystems as ts
from openmmtools_amd.system import system_to_desc

pytestmark = pytest.mark.gpu
KB = 0.008314462618153242


def _planes(n, p, seed):
    rng = np.random.default_rng(seed)
    a = rng.normal(size=(p, n, n)) + 1j * rng.normal(size=(p, n, n))
    a[0] *= 1e-6                      # per-plane scaling
    if p > 1:
        a[1] *= 3e4
    if p > 2:                         # mesh-like: a few spikes on an empty plane
        a[2] = 0.0
        a[2, rng.integers(0, n, 40), rng.integers(0, n, 40)] = rng.normal(size=40)
    return a.astype(np.complex64)


@pytest.mark.parametrize('n,p', [(75, 5), (80, 2), (64, 3), (30, 1)])
def test_forward_dft_matches_numpy(hip_engine_factory, n, p):
    eng = hip_engine_factory()
    a = _planes(n, p, 7 * n + p)
    got = eng.test_xy_mfma(a, mode=1)
    ref = np.fft.fft2(a.astype(np.complex128), axes=(1, 2))
    for k in range(p):
        assert np.abs(got[k] - ref[k]).max() < 1e-6 * np.abs(ref[k]).max(), (n, k)


@pytest.mark.parametrize('n,p', [(75, 4), (48, 3)])
def test_forward_inverse_round_trip(hip_engine_factory, n, p):
    eng = hip_engine_factory()
    a = _planes(n, p, 11 * n + p)
    back = eng.test_xy_mfma(a, mode=0) / float(n * n)
    for k in range(p):
        assert np.abs(back[k] - a[k]).max() < 1e-6 * np.abs(a[k]).max(), (n, k)


def test_mesh_forces_and_energy_match_the_fft_kernels(hip_engine_factory, monkeypatch):
    al = ts.AlanineDipeptideExplicit()
    box = np.diag(al.system.getDefaultPeriodicBoxVectors())
    rng = np.random.default_rng(5)
    R = 3
    x = np.stack([al.positions + 0.003 * rng.normal(size=al.positions.shape) for _ in range(R)])
    out = []
    for flag in ('1', '0'):
        monkeypatch.setenv('REMD_PME_XY_MFMA', flag)
        eng = hip_engine_factory()
        eng.set_system(system_to_desc(al.system)); eng.set_states(np.full(R, 1.0 / (KB * 300.0)))
        eng.set_integrator('V R O R V', 0.001, 1.0, 1, True, 1e-8)
        eng.set_replicas(R, 0, x, None, np.tile(box, (R, 1)), np.arange(R))
        comp = eng.energy_components()
        out.append((eng.get_forces(), np.array([c['pme_reciprocal'] for c in comp])))
    (f1, e1), (f0, e0) = out
    assert np.allclose(e1, e0, rtol=2e-6), (e1, e0)
    assert np.abs(f1 - f0).max() < 1e-5 * np.abs(f0).max()

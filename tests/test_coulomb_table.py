"""CPU-only: the Ewald direct-space force table of the force-only pair kernels (openmmtools_amd/csrc/coulomb_table.h).

libremd_hip.so's host-side hook evaluates the table exactly as the kernel does (same bins, same f32 Horner form); it is checked
against the f64 closed form (libremd_cpu.so's answer to the same hook and numpy/scipy here) in the unit that matters for a
force: the error relative to the bare Coulomb kernel 1/r^3 = u^-1.5, i.e. to q_i q_j k_e / r^2 — the Abramowitz & Stegun
7.1.26 erfc it replaces is good to 1.5e-7 in that unit."""
import ctypes as C
import os

import numpy as np
import pytest
from scipy.special import erfc

import oracle
from openmmtools_amd import _engine

CPU_LIB = os.path.join(os.path.dirname(os.path.abspath(oracle.__file__)), '_build', 'libremd_cpu.so')


def _hook(lib, alpha, rcc, u):
    u = np.ascontiguousarray(u, dtype=np.float32)
    out = np.empty_like(u)
    fp = C.POINTER(C.c_float)
    rc = lib.remd_test_coulomb_table(C.c_double(alpha), C.c_double(rcc), len(u), u.ctypes.data_as(fp), out.ctypes.data_as(fp))
    assert rc == 0
    return out


@pytest.mark.parametrize('rcc', [0.9, 1.0, 1.126, 1.21, 1.5])
def test_table_matches_the_closed_form(rcc):
    lib = _engine.load_library()                       # host arithmetic only: no device is touched
    alpha = np.sqrt(-np.log(2e-5)) / rcc
    rng = np.random.default_rng(7)
    u = np.exp(rng.uniform(np.log(2.0 ** -8), np.log(rcc * rcc), 400000)).astype(np.float32)
    u = np.concatenate([u, np.float32([2.0 ** -8, rcc * rcc]), np.float32(2.0) ** np.arange(-8, 1)])     # bin edges too
    u = u[u <= np.float32(rcc * rcc)]
    got = _hook(lib, alpha, rcc, u).astype(np.float64)
    ud = u.astype(np.float64)
    r = np.sqrt(ud)
    ref = -(erfc(alpha * r) / r + 2.0 * alpha / np.sqrt(np.pi) * np.exp(-alpha * alpha * ud)) / ud
    err = np.abs(got - ref) * ud ** 1.5
    assert err.max() < 2.5e-7, err.max()
    if not os.path.exists(CPU_LIB):
        oracle.build()
    cpu = _hook(_engine.load_library(CPU_LIB), alpha, rcc, u).astype(np.float64)
    assert np.abs(cpu - ref).max() <= 1.2e-7 * np.abs(ref).max()


def test_out_of_range_arguments_are_clamped_like_the_kernel_does():
    lib = _engine.load_library()
    alpha, rcc = 3.0, 1.0
    lo, hi = _hook(lib, alpha, rcc, np.float32([1e-6, 2.0 ** -8])), _hook(lib, alpha, rcc, np.float32([1.0, 7.5]))
    assert lo[0] == lo[1] and hi[0] == hi[1]

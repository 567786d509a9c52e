"""CPU oracle package — TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import anything from
here.  The product path (openmmtools_amd) never does; it fails loudly without libremd_hip.so.
"""
import ctypes as C
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, '_build', 'libmix_oracle.so')
_lib = None


def build():
    subprocess.check_call(['make', '-s', '-C', _HERE])


def mix_lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB):
            build()
        lib = C.CDLL(_LIB)
        dp, lp, ip, up = (C.POINTER(C.c_double), C.POINTER(C.c_int64), C.POINTER(C.c_int32), C.POINTER(C.c_uint32))
        lib.oracle_philox.argtypes = [up, up, up]
        lib.oracle_draw.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint64, up]
        lib.oracle_exp_det.argtypes = [C.c_double]
        lib.oracle_exp_det.restype = C.c_double
        lib.oracle_mix_swap_all.argtypes = [C.c_uint64, C.c_int64, C.c_int, C.c_int, dp, lp, lp, lp, C.c_int64]
        lib.oracle_mix_swap_sequence.argtypes = [C.c_int, C.c_int, dp, lp, lp, lp, C.c_int64, ip, ip, dp]
        lib.oracle_mix_swap_neighbors.argtypes = [C.c_uint64, C.c_int64, C.c_int, C.c_int, dp, lp, lp, lp]
        lib.oracle_sams_global_jump.argtypes = [C.c_uint64, C.c_int64, C.c_int, C.c_int, dp, dp, lp, lp, lp, dp]
        lib.oracle_sams_update_logZ.argtypes = [C.c_int, C.c_int, lp, dp, dp, C.c_double, C.c_int64, C.c_int,
                                                C.c_int64, C.c_int, dp, dp]
        _lib = lib
    return _lib


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


def philox(ctr, key):
    lib = mix_lib()
    c = np.array(ctr, dtype=np.uint32); k = np.array(key, dtype=np.uint32); o = np.zeros(4, dtype=np.uint32)
    lib.oracle_philox(_p(c, C.c_uint32), _p(k, C.c_uint32), _p(o, C.c_uint32))
    return o


def draw(seed, stream, a, b, t):
    o = np.zeros(4, dtype=np.uint32)
    mix_lib().oracle_draw(seed, stream, a, b, t, _p(o, C.c_uint32))
    return o


def exp_det(x):
    return mix_lib().oracle_exp_det(float(x))


def mix(scheme, seed, iteration, ukl, labels, log_weights=None, n_attempts=-1):
    """Sequential reference mixing.  Returns labels, n_accepted, n_proposed, log_P (SAMS) like HipEngine.mix_host."""
    lib = mix_lib()
    ukl = np.ascontiguousarray(ukl, dtype=np.float64)
    R, K = ukl.shape
    labels = np.ascontiguousarray(labels, dtype=np.int64).copy()
    nacc = np.zeros((K, K), dtype=np.int64); nprop = np.zeros((K, K), dtype=np.int64)
    dp, lp = C.c_double, C.c_int64
    logP = None
    if scheme == 'swap-all':
        lib.oracle_mix_swap_all(seed, iteration, R, K, _p(ukl, dp), _p(labels, lp), _p(nacc, lp), _p(nprop, lp), n_attempts)
    elif scheme == 'swap-neighbors':
        lib.oracle_mix_swap_neighbors(seed, iteration, R, K, _p(ukl, dp), _p(labels, lp), _p(nacc, lp), _p(nprop, lp))
    elif scheme == 'sams-global-jump':
        lw = np.ascontiguousarray(log_weights, dtype=np.float64)
        logP = np.zeros((R, K))
        lib.oracle_sams_global_jump(seed, iteration, R, K, _p(ukl, dp), _p(lw, dp), _p(labels, lp), _p(nacc, lp),
                                    _p(nprop, lp), _p(logP, dp))
    elif scheme in (None, 'none'):
        pass
    else:
        raise ValueError(scheme)
    return labels, nacc, nprop, logP


def mix_sequence(ukl, labels, ii, jj, uu):
    lib = mix_lib()
    ukl = np.ascontiguousarray(ukl, dtype=np.float64)
    R, K = ukl.shape
    labels = np.ascontiguousarray(labels, dtype=np.int64).copy()
    nacc = np.zeros((K, K), dtype=np.int64); nprop = np.zeros((K, K), dtype=np.int64)
    ii = np.ascontiguousarray(ii, dtype=np.int32); jj = np.ascontiguousarray(jj, dtype=np.int32)
    uu = np.ascontiguousarray(uu, dtype=np.float64)
    lib.oracle_mix_swap_sequence(R, K, _p(ukl, C.c_double), _p(labels, C.c_int64), _p(nacc, C.c_int64),
                                 _p(nprop, C.c_int64), len(ii), _p(ii, C.c_int32), _p(jj, C.c_int32), _p(uu, C.c_double))
    return labels, nacc, nprop

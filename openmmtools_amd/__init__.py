"""openmmtools_amd: MI355X-native replica-exchange hot path behind the openmmtools multistate API."""
import os as _os
# HIP maps streams onto GPU_MAX_HW_QUEUES hardware queues per priority (default 4).  A handle that propagates its replicas as two phases
# (include/remd_hip.h: remd_set_phases) uses two streams of normal and two of raised priority; with the default, a fifth queue appears next
# to the process's default stream, and hardware queues beyond the chip's four pipes are time-sliced (measured: two phases 55 % slower than
# one instead of 10 % faster, profiles/r06_phases_hw_queues.txt).  Two queues per priority keep the process at four.  Read by the HIP
# runtime when it starts, so it has to be in the environment before the first HIP call of the process; a value the user set is kept.
import sys as _sys
if 'GPU_MAX_HW_QUEUES' not in _os.environ:
    _t = _sys.modules.get('torch')
    if _t is not None and getattr(_t, 'cuda', None) is not None and _t.cuda.is_initialized():
        # the HIP runtime is up already and has read its flags: the queue limit can no longer reach it -- keep the engines at one block
        # (libremd_hip.so would otherwise see the variable and run two phases on five hardware queues)
        _os.environ.setdefault('REMD_PHASES', '1')
    else:
        _os.environ['GPU_MAX_HW_QUEUES'] = '2'
from . import unit, constants, system, states, integrators, mcmc, cache   # noqa: F401
__version__ = '0.1.0'

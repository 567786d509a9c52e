"""openmmtools_amd: MI355X-native replica-exchange hot path behind the openmmtools multistate API."""
import os as _os
# HIP maps streams onto GPU_MAX_HW_QUEUES hardware queues per priority (default 4).  A handle that propagates its replicas as two phases
# (include/remd_hip.h: remd_set_phases) uses two streams of normal and two of raised priority; with the default, a fifth queue appears next
# to the process's default stream, and hardware queues beyond the chip's four pipes are time-sliced (measured: two phases 55 % slower than
# one instead of 10 % faster, profiles/r06_phases_hw_queues.txt).  Two queues per priority keep the process at four.  Read by the HIP
# runtime when it starts, so it has to be in the environment before the first HIP call of the process; a value the user set is kept.
_os.environ.setdefault('GPU_MAX_HW_QUEUES', '2')
from . import unit, constants, system, states, integrators, mcmc, cache   # noqa: F401
__version__ = '0.1.0'

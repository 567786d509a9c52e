"""openmmtools_amd: MI355X-native replica-exchange hot path behind the openmmtools multistate API."""
from . import unit, constants, system, states, integrators, mcmc, cache   # noqa: F401
__version__ = '0.1.0'

"""ParallelTemperingSampler: temperature ladder + O(N) reduced-potential matrix.

Mirrors openmmtools/multistate/paralleltempering.py (class :44): ``create`` (:109-173) builds
exponentially spaced temperatures (np.logspace, :162); the O(N) energy shortcut
``u_kl = beta_l * U_k`` (:175-215) is what the device's assemble_ukl kernel evaluates for
every replica from one potential-energy pass.
"""
import copy
import numpy as np
from .. import states
from .replicaexchange import ReplicaExchangeSampler


class ParallelTemperingSampler(ReplicaExchangeSampler):
    _TITLE_TEMPLATE = 'Parallel tempering simulation created using ParallelTempering class of openmmtools_amd.multistate on {}'

    def create(self, thermodynamic_state, sampler_states, storage=None, min_temperature=None, max_temperature=None,
               n_temperatures=None, temperatures=None, **kwargs):
        if not isinstance(thermodynamic_state, states.ThermodynamicState):
            raise ValueError("ParallelTempering only accepts a single ThermodynamicState!\n"
                             "If you have already set temperatures in your list of states, please use the "
                             "standard ReplicaExchange class with your list of states.")
        if temperatures is not None:
            temperatures = [float(t) for t in temperatures]
            n_temperatures = len(temperatures)
        elif min_temperature is not None and max_temperature is not None and n_temperatures is not None:
            temperatures = np.logspace(np.log10(float(min_temperature)), np.log10(float(max_temperature)),
                                       num=n_temperatures)                                   # :162
        else:
            raise ValueError("Either 'temperatures' or ('min_temperature', 'max_temperature', "
                             "and 'n_temperatures') must be provided.")
        thermodynamic_states = [copy.deepcopy(thermodynamic_state) for _ in range(n_temperatures)]
        for state, temperature in zip(thermodynamic_states, temperatures):
            state.temperature = temperature
        super().create(thermodynamic_states, sampler_states, storage=storage, **kwargs)

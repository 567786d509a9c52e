"""ReplicaExchangeSampler: Gibbs state-label mixing on the device.

Mirrors openmmtools/multistate/replicaexchange.py (class :52): ``replica_mixing_scheme``
validation (:220-234), round-robin tiling of sampler states (:239-253) and ``_mix_replicas``
(:255-292), whose swap-all inner loop (_mix_all_replicas_numba :294-349) and neighbour scheme
(:366-380) run as HIP kernels (csrc/mix.hip) instead of numba / Python.
"""
import numpy as np
from .multistatesampler import MultiStateSampler
from .comm import SingleProcessComm
from ..utils import time_it


class ReplicaExchangeSampler(MultiStateSampler):
    _TITLE_TEMPLATE = 'Replica-exchange sampler simulation created using ReplicaExchangeSampler class of openmmtools_amd.multistate on {}'

    def __init__(self, replica_mixing_scheme='swap-all', **kwargs):
        super().__init__(**kwargs)
        self.replica_mixing_scheme = replica_mixing_scheme

    def _ctor_kwargs(self):
        return dict(replica_mixing_scheme=self._replica_mixing_scheme)

    @property
    def replica_mixing_scheme(self):
        return self._replica_mixing_scheme

    @replica_mixing_scheme.setter
    def replica_mixing_scheme(self, scheme):
        """replicaexchange.py:220-234."""
        supported = ['swap-all', 'swap-neighbors', None]
        if scheme not in supported:
            raise ValueError("Unknown replica mixing scheme '{}'. Supported values are {}.".format(scheme, supported))
        if getattr(self, 'locality', None) is not None and scheme != 'swap-neighbors':
            raise ValueError("replica_mixing_scheme must be 'swap-neighbors' if locality is used")
        self._replica_mixing_scheme = scheme

    def _pre_write_create(self, thermodynamic_states, sampler_states, *args, **kwargs):
        """replicaexchange.py:239-253: one replica per state, sampler states tiled round-robin."""
        n_states = len(thermodynamic_states)
        if len(sampler_states) > n_states:
            raise ValueError('Passed {} SamplerStates but only {} ThermodynamicStates'.format(
                len(sampler_states), n_states))
        sampler_states = [sampler_states[i % len(sampler_states)] for i in range(n_states)]
        super()._pre_write_create(thermodynamic_states, sampler_states, *args, **kwargs)

    def _mix_replicas(self, rng_iteration=None):
        """replicaexchange.py:255-292."""
        it = self._iteration if rng_iteration is None else rng_iteration
        K = self.n_states
        if self._replica_mixing_scheme is None:
            self._n_accepted_matrix[:, :] = 0
            self._n_proposed_matrix[:, :] = 0
            return self._replica_thermodynamic_states
        with time_it('Mixing of replicas'):                                   # replicaexchange.py:265
            labels, nacc, nprop = self._device_mix(self._replica_mixing_scheme, it)
        self._n_accepted_matrix[:, :] = nacc[:K, :K]
        self._n_proposed_matrix[:, :] = nprop[:K, :K]
        n_prop = self._n_proposed_matrix.sum()
        self._swap_fraction_accepted = float(self._n_accepted_matrix.sum()) / n_prop if n_prop > 0 else 0.0
        return labels

    def _device_mix(self, scheme, it, log_weights=None):
        eng = self._engine
        labels_in = self._replica_thermodynamic_states
        K = self.n_states
        distributed = not isinstance(self._comm, SingleProcessComm)
        if getattr(self, '_mix_from_stored_energies', False):
            # first mix after from_storage: the reference mixes with the energies read back from storage
            # (multistatesampler.py:1003-1020), not with energies recomputed from the f4 checkpoint positions
            self._mix_from_stored_energies = False
            out = eng.mix_host(scheme, it, np.ascontiguousarray(self._energy_thermodynamic_states[:, :K]), labels_in,
                               log_weights=log_weights)
        elif distributed and getattr(eng, 'is_device', False) and self._device_ukl is not None:
            out = eng.mix(scheme, it, labels_in, d_ukl=self._device_ukl.data_ptr(), R=self.n_replicas, K=K,
                          ld=self._K_total, log_weights=log_weights)
        elif distributed:
            out = eng.mix_host(scheme, it, self._host_ukl_full[:, :K], labels_in, log_weights=log_weights)
        else:
            out = eng.mix(scheme, it, labels_in, R=self.n_replicas, K=K, ld=self._K_total, log_weights=log_weights)
        labels, nacc, nprop, logP = out
        if self.verify_labels and self._comm.world_size > 1:
            ref = self._comm.broadcast_labels(labels)
            if not np.array_equal(ref, labels):
                raise RuntimeError('replicated mixing diverged between ranks')
        self._last_log_P = logP
        return labels, nacc, nprop

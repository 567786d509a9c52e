from .multistatesampler import MultiStateSampler            # noqa: F401
from .replicaexchange import ReplicaExchangeSampler          # noqa: F401
from .paralleltempering import ParallelTemperingSampler      # noqa: F401
from .sams import SAMSSampler                                # noqa: F401
from .utils import SimulationNaNError                        # noqa: F401
from .multistatereporter import MultiStateReporter          # noqa: F401
from .analysis import (MultiStateSamplerAnalyzer, MultiPhaseAnalyzer, ReplicaExchangeAnalyzer, ParallelTemperingAnalyzer, SAMSAnalyzer, MBAR,   # noqa: F401
                       get_decorrelation_time, get_equilibration_data, get_equilibration_data_per_sample,
                       remove_unequilibrated_data, subsample_data_along_axis, generate_phase_name)

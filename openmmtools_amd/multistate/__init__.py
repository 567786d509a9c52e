from .multistatesampler import MultiStateSampler            # noqa: F401
from .replicaexchange import ReplicaExchangeSampler          # noqa: F401
from .paralleltempering import ParallelTemperingSampler      # noqa: F401
from .sams import SAMSSampler                                # noqa: F401
from .utils import SimulationNaNError                        # noqa: F401
from .multistatereporter import MultiStateReporter          # noqa: F401
from .analysis import MultiStateSamplerAnalyzer, MBAR      # noqa: F401

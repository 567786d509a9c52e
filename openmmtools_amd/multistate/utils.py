"""multistate/utils.py:51 SimulationNaNError and small helpers."""


class SimulationNaNError(Exception):
    """Raised when a replica's positions, velocities or reduced potential become NaN."""


def __getattr__(name):
    """The time-series helpers of multistate/utils.py:60-300 live next to the analyzer (analysis.py); resolved lazily so that
    ``multistate.utils`` stays importable from the sampler without pulling the analysis code in."""
    if name in ('generate_phase_name', 'get_decorrelation_time', 'get_equilibration_data_per_sample', 'get_equilibration_data',
                'remove_unequilibrated_data', 'subsample_data_along_axis'):
        from . import analysis
        return getattr(analysis, name)
    raise AttributeError(name)

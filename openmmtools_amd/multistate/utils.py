"""multistate/utils.py:51 SimulationNaNError and small helpers."""


class SimulationNaNError(Exception):
    """Raised when a replica's positions, velocities or reduced potential become NaN."""

"""Wall-clock tracing helpers with the reference's names (openmmtools/utils/utils.py:65-183: ``Timer``, ``time_it``,
``with_timer``), which the multistate sampler hangs on its per-iteration phases (decorators on
``_propagate_replicas`` :1287, ``_compute_energies`` :1436, ``_report_iteration`` :1191, ``minimize`` :611, the
analysis :1525/:1624 and ``time_it('Mixing of replicas')`` replicaexchange.py:265).  Host wall time only; device-side
per-kernel-class timing is the engine's profile scopes (``HipEngine.profile_enable`` / ``profile_get``)."""
import contextlib
import functools
import logging
import time

logger = logging.getLogger(__name__)


class Timer:
    """Named stopwatches.  ``stop`` returns the elapsed seconds; ``report_timing`` logs what finished at debug level."""

    def __init__(self):
        self._started, self._finished = {}, {}

    def reset_timing_statistics(self, benchmark_id=None):
        if benchmark_id is None:
            self._started.clear(); self._finished.clear()
        else:
            self._started.pop(benchmark_id, None); self._finished.pop(benchmark_id, None)

    def start(self, benchmark_id):
        self._started[benchmark_id] = time.perf_counter()

    def partial(self, benchmark_id):
        if benchmark_id not in self._started:
            logger.warning("Couldn't return partial timing for %s", benchmark_id)
            return None
        return time.perf_counter() - self._started[benchmark_id]

    def stop(self, benchmark_id):
        if benchmark_id not in self._started:
            logger.warning("Can't stop timing for %s", benchmark_id)
            return None
        elapsed = time.perf_counter() - self._started[benchmark_id]
        self._finished[benchmark_id] = elapsed
        return elapsed

    def report_timing(self, clear=True):
        out = dict(self._finished)
        for name, seconds in out.items():
            logger.debug('%s took %8.3fs', name, seconds)
        if clear:
            self.reset_timing_statistics()
        return out


@contextlib.contextmanager
def time_it(task_name):
    """Log the wall time of a block at debug level."""
    timer = Timer()
    timer.start(task_name)
    try:
        yield timer
    finally:
        timer.stop(task_name)
        timer.report_timing()


def with_timer(task_name):
    """Decorator form of :func:`time_it`."""
    def decorate(func):
        @functools.wraps(func)
        def wrapper(*args, **kwargs):
            with time_it(task_name):
                return func(*args, **kwargs)
        return wrapper
    return decorate

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


# ---- serialization (openmmtools/utils/utils.py:606-690) -------------------------------------------------------------------------
_SERIALIZED_MANGLED_PREFIX = '_serialized__'


def serialize(instance, **kwargs):
    """The dictionary of ``instance.__getstate__(**kwargs)`` plus the module and class names it is rebuilt from (:611-646)."""
    try:
        serialization = dict(instance.__getstate__(**kwargs))
    except AttributeError:
        raise ValueError('Cannot serialize class {} without a __getstate__ method'.format(instance.__class__.__name__))
    serialization[_SERIALIZED_MANGLED_PREFIX + 'module_name'] = instance.__module__
    serialization[_SERIALIZED_MANGLED_PREFIX + 'class_name'] = instance.__class__.__name__
    return serialization


def deserialize(serialization):
    """The instance a ``serialize`` dictionary describes: its class is looked up by name (classes of this package only) and
    filled through ``__setstate__`` without running ``__init__`` (:649-686)."""
    import importlib
    serialization = dict(serialization)
    names = []
    for key in ('module_name', 'class_name'):
        try:
            names.append(serialization.pop(_SERIALIZED_MANGLED_PREFIX + key))
        except KeyError:
            raise ValueError('Cannot find {} in the serialization. Was the original object serialized with '
                             'openmmtools.utils.serialize()?'.format(key))
    module_name, class_name = names
    if module_name.startswith('openmmtools.'):                       # a dictionary written by the reference: same class, this package
        module_name = 'openmmtools_amd.' + module_name[len('openmmtools.'):]
    if not (module_name == 'openmmtools_amd' or module_name.startswith('openmmtools_amd.')):
        raise ValueError('refusing to deserialize a class outside this package: {}.{}'.format(module_name, class_name))
    cls = getattr(importlib.import_module(module_name), class_name)
    instance = cls.__new__(cls)
    try:
        instance.__setstate__(serialization)
    except AttributeError:
        raise ValueError('Cannot deserialize class {} without a __setstate__ method'.format(class_name))
    return instance

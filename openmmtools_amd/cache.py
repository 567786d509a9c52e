"""openmmtools/cache.py stand-in: the names scripts written for the reference configure.

The reference keeps OpenMM Contexts in an LRU ``ContextCache`` (cache.py:200-470; ``global_context_cache`` :560) and the samplers
propagate / evaluate through ``sampler_context_cache`` / ``energy_context_cache`` (multistatesampler.py:1755-1764).  Here the
pool of device state is ONE engine handle per GPU (``_engine.HipEngine``: every replica of the rank in fused device arrays), so
there is nothing to cache; this module keeps the configuration surface -- capacity, time to live, platform -- so that code which
sets it keeps running, and translates the one setting that still means something: a platform's device index
(``platform_properties['DeviceIndex']``) is the engine's GPU.
"""


class ContextCache:
    """cache.py:200-470 (configuration only: no Contexts exist here)."""

    def __init__(self, platform=None, platform_properties=None, capacity=128, time_to_live=None, **kwargs):
        if platform_properties is not None and platform is None:
            raise ValueError('To set platform_properties, you need to also specify the platform.')      # cache.py:268-270
        self._platform = platform
        self._platform_properties = platform_properties
        self.capacity = capacity
        self.time_to_live = time_to_live

    @property
    def platform(self):
        return self._platform

    @platform.setter
    def platform(self, new_platform):
        """cache.py:286-293: the platform can change only while the cache is empty -- it always is."""
        self._platform = new_platform

    @property
    def device_index(self):
        """The GPU the engine of a sampler configured with this cache should use (``DeviceIndex`` / ``CudaDeviceIndex``)."""
        props = self._platform_properties or {}
        for key in ('DeviceIndex', 'CudaDeviceIndex', 'OpenCLDeviceIndex', 'HipDeviceIndex'):
            if key in props:
                return int(str(props[key]).split(',')[0])
        return 0

    def make_engine(self):
        """An engine handle on this cache's device (what ``get_context`` amounts to here)."""
        from ._engine import HipEngine
        return HipEngine(device=self.device_index)

    def empty(self):
        """cache.py:317-319."""

    def __len__(self):
        return 0

    def __getstate__(self):
        return dict(capacity=self.capacity, time_to_live=self.time_to_live, platform=self._platform,
                    platform_properties=self._platform_properties)

    def __setstate__(self, state):
        self.__init__(**state)


class DummyContextCache(ContextCache):
    """cache.py:473-556: the cache that never keeps a Context."""

    def __init__(self, platform=None, platform_properties=None, **kwargs):
        super().__init__(platform=platform, platform_properties=platform_properties, capacity=0, time_to_live=None)


global_context_cache = ContextCache(capacity=None, time_to_live=None)       # cache.py:560

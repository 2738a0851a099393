"""MI355X-native renderer for raytracer_cpp's per-pixel trace loop.

The product is the C-ABI shared library ``librt_amd.so`` (include/rt_api.h, sources under
``raytracer_amd/csrc``); ``raytracer_amd.capi`` binds it with ctypes for tests and bench.py.
"""
from . import capi  # noqa: F401  (fails loudly if librt_amd.so is missing)

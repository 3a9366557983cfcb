"""raider_amd - MI355X-native engine for RAiDER's tropospheric-delay hot path.

Public surface mirrors the reference modules for that path (tools/RAiDER/{delay,delayFcns,losreader,
utilFcns,interpolator}.py and the `interpolate` / `makePoints` extensions); all arithmetic runs in
hand-written HIP kernels behind the C ABI of include/raider_hip.h.  No CPU fallback exists.
"""
__version__ = '0.1.0'

from ._lib import Context, NoLevels, load as load_library  # noqa: F401
from .engine import Cube, Rays, nparts_from_maxlen  # noqa: F401

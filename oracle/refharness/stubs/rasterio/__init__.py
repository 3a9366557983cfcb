"""Build-owned import-only stand-in for `rasterio` (oracle harness only)."""
from . import crs, transform, drivers  # noqa: F401


def open(*a, **k):
    raise OSError('stub rasterio cannot open files')

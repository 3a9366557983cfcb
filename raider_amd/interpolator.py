"""scipy-style callable over the GPU `interpolate` (counterpart of tools/RAiDER/interpolator.py:19-69).

Edge rule = the native extension's, not scipy's: with a fill value, a query lying ON the last node of an axis is
filled (interpolate.h:23-38,58-65); without one, queries outside the grid are linearly extrapolated."""
import numpy as np

from .interpolate import interpolate


class RegularGridInterpolator:
    def __init__(self, grid, values, fill_value=None, assume_sorted=False, max_threads=8):
        self.grid, self.values = grid, values
        self.fill_value, self.assume_sorted, self.max_threads = fill_value, assume_sorted, max_threads

    def __call__(self, points):
        """points: an (..., ndim) array, or a tuple of equally shaped coordinate arrays (one per axis)."""
        if isinstance(points, tuple):
            shapes = {np.shape(p) for p in points}
            assert len(shapes) == 1, 'All dimensions must contain the same number of points!'
            points = np.stack(points, axis=-1)
        points = np.asarray(points)
        lead = points.shape[:-1]
        flat = interpolate(self.grid, self.values, points.reshape(-1, points.shape[-1]), fill_value=self.fill_value,
                           assume_sorted=self.assume_sorted, max_threads=self.max_threads)
        return flat.reshape(lead)

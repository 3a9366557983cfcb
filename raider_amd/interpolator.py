"""scipy-like wrapper over the GPU `interpolate` (tools/RAiDER/interpolator.py:19-69)."""
import numpy as np

from .interpolate import interpolate


class RegularGridInterpolator:
    """interpolator.py:19-69 - note the native edge rule (a query ON the last grid node is filled)."""

    def __init__(self, grid, values, fill_value=None, assume_sorted=False, max_threads=8):
        self.grid = grid
        self.values = values
        self.fill_value = fill_value
        self.assume_sorted = assume_sorted
        self.max_threads = max_threads

    def __call__(self, points):
        if isinstance(points, tuple):
            shape = points[0].shape
            for arr in points:
                assert arr.shape == shape, 'All dimensions must contain the same number of points!'
            interp_points = np.stack(points, axis=-1)
            in_shape = interp_points.shape
            interp_points = interp_points.reshape(-1, in_shape[-1])
        elif points.ndim > 2:
            in_shape = points.shape
            interp_points = points.reshape((int(np.prod(points.shape[:-1])),) + (points.shape[-1],))
        else:
            interp_points = points
            in_shape = interp_points.shape
        out = interpolate(self.grid, self.values, interp_points, fill_value=self.fill_value,
                          assume_sorted=self.assume_sorted, max_threads=self.max_threads)
        return out.reshape(in_shape[:-1])

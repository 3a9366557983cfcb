"""Weather-model cube producer on the GPU.

Mirrors the processing chain of the reference's WeatherModel.load (tools/RAiDER/models/weatherModel.py:235-262):
``_find_e`` -> ``_uniform_in_z`` -> ``_checkForNans`` -> wet / hydrostatic refractivity -> ``_adjust_grid`` -> ``_getZTD``,
but lands the result directly in the two device cubes the delay kernels read (no NetCDF round trip through
``write()`` / ``getInterpolators``).  Reading GRIB/NetCDF model files and the geopotential -> height conversion stay with
the caller (they are I/O, not part of this path).
"""
import ctypes as C

import numpy as np

from . import _lib as L
from ._lib import Context, check
from .engine import Cube, _is_dev, f64, ptr

# models/weatherModel.py:25-31 (the TOTAL_LEVELS_IN_MODEL heights every model is resampled to)
_ZMIN = -100.0
_ZREF = 80000.0
MODEL_LEVEL_HEIGHTS = np.array(
    [-100, 0, 50, 100, 150, 200, 250, 300, 350, 400, 450, 500, 600, 700, 800, 900, 1000, 1100, 1200, 1300, 1400, 1500, 1600,
     1700, 1800, 1900, 2000, 2100, 2200, 2300, 2400, 2500, 2600, 2700, 2800, 2900, 3000, 3100, 3200, 3300, 3400, 3500, 3600,
     3700, 3800, 3900, 4000, 4200, 4400, 4600, 4800, 5000, 5250, 5500, 5750, 6000, 6250, 6500, 6750, 7000, 7500, 8000, 8500,
     9000, 9500, 10000, 11000, 12000, 13000, 14000, 15000, 16000, 17000, 18000, 19000, 20000, 25000, 30000, 35000, 40000],
    dtype=np.float64)


class ProcessedModel:
    """What WeatherModel.load leaves behind, GPU-resident: ``pointwise`` = (wet, hydro) refractivity cube (f32),
    ``total`` = (wet_total, hydro_total) zenith-delay cube (f64); ``t``, ``p``, ``e`` (f32, (y, x, z)) when asked for."""

    def __init__(self, pointwise, total, zs, t=None, p=None, e=None, proj=4326):
        self.pointwise, self.total, self.zs = pointwise, total, zs
        self.t, self.p, self.e = t, p, e
        self.proj = proj            # CRS of the x/y axes, as tropo_delay reads it off the model file (delay.py:66-73)

    # mapping view with the processed file's variable names (weatherModel.py:685-693; fields in file order (z, y, x)), so an
    # instance can be handed to tropo_delay / getInterpolators wherever they take a weather-model file
    _KEYS = ('x', 'y', 'z', 'wet', 'hydro', 'wet_total', 'hydro_total', 'proj')

    def keys(self):
        return self._KEYS

    def __contains__(self, k):
        return k in self._KEYS

    def __getitem__(self, k):
        if k == 'x': return self.pointwise.grid[1]
        if k == 'y': return self.pointwise.grid[0]
        if k == 'z': return self.pointwise.grid[2]
        if k == 'proj': return self.proj
        if k in ('wet', 'hydro'): return np.ascontiguousarray(self.pointwise.read()[k == 'hydro'].transpose(2, 0, 1))
        if k in ('wet_total', 'hydro_total'): return np.ascontiguousarray(self.total.read()[k == 'hydro_total'].transpose(2, 0, 1))
        raise KeyError(k)

    def interpolators(self, kind='pointwise'):
        """getInterpolators(wm_file, kind) (delayFcns.py:23-58) without the file"""
        from .delayFcns import interpolators_from_cube
        return interpolators_from_cube(self.pointwise if kind == 'pointwise' else self.total)


def cubes_from_model_levels(xs, ys, zs, p, t, hum, humidity_type='q', new_z=None, k1=0.776, k2=0.233, k3=3.75e3,
                            zmin=_ZMIN, return_state=False, ctx=None):
    """zs, p, t, hum: (ny, nx, nlev) model-level heights (m, ascending in the last axis), pressure (Pa), temperature (K)
    and specific ('q') or relative ('rh', %) humidity; xs, ys: the cube's ascending 1-D axes; new_z: output levels
    (default: the reference's fixed 145-level table is model specific; here the caller passes it, or gets
    MODEL_LEVEL_HEIGHTS clipped below the column tops like _uniform_in_z does with _zlevels)."""
    if humidity_type not in ('q', 'rh'):
        raise RuntimeError('Not a valid humidity type')        # weatherModel.py:340-341
    ctx = ctx or Context.default()
    xs, ys = f64(xs), f64(ys)
    dev = _is_dev(zs)
    if dev:
        import torch
        arrs = [a.to(torch.float64).contiguous() for a in (zs, p, t, hum)]
        ctx.adopt_torch_stream(arrs[0])
        shape = tuple(arrs[0].shape)
    else:
        arrs = [f64(a) for a in (zs, p, t, hum)]
        shape = arrs[0].shape
    if len(shape) != 3 or shape[:2] != (ys.size, xs.size) or any(tuple(a.shape) != tuple(shape) for a in arrs):
        raise ValueError(f'model-level arrays must all be (ny, nx, nlev) = ({ys.size}, {xs.size}, nlev); got {shape}')
    if new_z is None:
        new_z = MODEL_LEVEL_HEIGHTS
    new_z = f64(new_z)
    nzo = new_z.size + (1 if zmin < new_z[0] else 0)
    state = [None, None, None]
    if return_state:
        if dev:
            import torch
            state = [torch.empty(shape[:2] + (nzo,), dtype=torch.float32, device=arrs[0].device) for _ in range(3)]
        else:
            state = [np.empty(shape[:2] + (nzo,), np.float32) for _ in range(3)]
    hp, ht = C.c_void_p(), C.c_void_p()
    check(ctx.lib.rdr_cubes_from_model_levels(
        ctx.handle, ptr(ys), ys.size, ptr(xs), xs.size, ptr(arrs[0]), ptr(arrs[1]), ptr(arrs[2]), ptr(arrs[3]),
        0 if humidity_type == 'q' else 1, shape[2], ptr(new_z), new_z.size, float(k1), float(k2), float(k3), float(zmin),
        L.RDR_DEVICE if dev else L.RDR_HOST, C.byref(hp), C.byref(ht),
        ptr(state[0]) if return_state else None, ptr(state[1]) if return_state else None, ptr(state[2]) if return_state else None),
        ctx.handle)
    pw, tot = Cube._from_handle(ctx, hp), Cube._from_handle(ctx, ht)
    return ProcessedModel(pw, tot, pw.grid[2].copy(), *state)

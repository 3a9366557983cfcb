"""Azimuth-time-grid temporal weighting on the GPU (mirror of tools/RAiDER/s1_azimuth_timing.py).

The reference interpolates 2-3 hourly weather cubes to the per-voxel Sentinel-1 acquisition time:
  get_azimuth_time_grid (:90-147)          zero-Doppler time of every (lon, lat, h) voxel + the range delay  -> datetime64[ms]
  get_inverse_weights_for_dates (:326-399) normalised inverse-|dt| weights of the model times, masked to one model step
  combine_weather_files (cli/raider.py:791-832)   ds_out[var] = sum(w_i * ds_i[var])
Here the geometry solve, the weights and the weighted combination all run on the device; the date bookkeeping
(get_n_closest_datetimes, get_times_for_azimuth_interpolation) is host logic.  The SLC / orbit-file lookup of
get_s1_azimuth_time_grid (:150-216) needs the network (asf_search, s1_orbits) and is out of scope: pass an Orbit.
"""
import ctypes as C
import datetime as dt

import numpy as np

from . import _lib as L
from ._lib import Context, check, f64, ptr
from .engine import Cube
from .orbits import Orbit

SPEED_OF_LIGHT = 299792458.0     # isce3.core.speed_of_light


def get_n_closest_datetimes(ref_time, n_target_times, time_step_hours):
    """s1_azimuth_timing.py:204-266: the n model times (multiples of the step since midnight) closest to ref_time, ordered by
    distance; of two equally distant ones the earlier comes first."""
    if (24 % time_step_hours) != 0:
        raise ValueError('The time step does not evenly divide 24 hours;Time step has period > 1 day and depends when model starts')
    step = dt.timedelta(hours=time_step_hours)
    midnight = dt.datetime(ref_time.year, ref_time.month, ref_time.day)

    def down(t):
        return midnight + ((t - midnight) // step) * step

    def up(t):
        lo = down(t)
        return lo if lo == t else lo + step

    candidates = []
    for k in range(int(np.ceil(n_target_times / 2))):
        candidates.extend({down(ref_time - k * step), up(ref_time + k * step)})
    candidates.sort(key=lambda t: (abs(ref_time - t), t))
    return candidates[:n_target_times]


def get_times_for_azimuth_interpolation(ref_time, time_step_hours, buffer_in_seconds=300):
    """s1_azimuth_timing.py:269-323: 2 or 3 closest model times within one time step (plus the buffer) of ref_time."""
    limit = time_step_hours * 60 * 60 + buffer_in_seconds
    return [t for t in get_n_closest_datetimes(ref_time, 3, time_step_hours) if abs((ref_time - t).total_seconds()) < limit]


def _seconds(times, epoch):
    """datetime64 array / datetimes -> float64 seconds since `epoch` (a datetime); exact for ms / us resolution inputs"""
    e64 = np.datetime64(epoch, 'us')
    if isinstance(times, np.ndarray) and np.issubdtype(times.dtype, np.datetime64):
        return (times.astype('datetime64[us]') - e64).astype(np.int64) / 1e6
    return np.array([(t - epoch).total_seconds() for t in times], dtype=np.float64)


def get_inverse_weights_for_dates(azimuth_time_array, dates, inverse_regularizer=1e-9, temporal_window_hours=None, ctx=None):
    """s1_azimuth_timing.py:326-399.  azimuth_time_array: np.datetime64 array (any shape), or float64 seconds relative to
    dates[0] (NumPy or a torch tensor on the GPU).  Returns a list of weight arrays, one per date."""
    if len(set(dates)) != len(dates):
        raise ValueError('Dates provided must be unique')
    if len(dates) == 0:
        raise ValueError('No dates provided')
    if not all(isinstance(d, dt.datetime) for d in dates):
        raise TypeError('dates must be all datetimes')
    ctx = ctx or Context.default()
    epoch = dates[0]
    d_s = f64(_seconds(dates, epoch))
    window = -1.0 if temporal_window_hours is None else float(temporal_window_hours) * 3600.0
    if len(dates) == 1 and temporal_window_hours is None:
        raise ValueError('min() arg is an empty sequence')        # what the reference's window inference raises
    dev = hasattr(azimuth_time_array, 'data_ptr')
    if dev:
        import torch
        az = azimuth_time_array.contiguous()
        ctx.adopt_torch_stream(az)
        shape = tuple(az.shape); n = az.numel()
        out = torch.empty((len(dates),) + shape, dtype=torch.float64, device=az.device)
    else:
        a = np.asarray(azimuth_time_array)
        az = f64(_seconds(a, epoch) if np.issubdtype(a.dtype, np.datetime64) else a)
        shape = az.shape; n = az.size
        out = np.empty((len(dates),) + shape)
    check(ctx.lib.rdr_inverse_time_weights(ctx.handle, ptr(az), n, ptr(d_s), len(dates), window, float(inverse_regularizer), ptr(out),
                                           L.RDR_DEVICE if dev else L.RDR_HOST), ctx.handle)
    return [out[i] for i in range(len(dates))]


def get_azimuth_time_grid(lon_mesh, lat_mesh, hgt_mesh, orb, as_datetime64=True, ctx=None):
    """s1_azimuth_timing.py:90-147 for an `raider_amd.orbits.Orbit`: zero-Doppler azimuth time of every voxel plus the one-way
    range delay (:138-139), truncated to milliseconds.  Returns datetime64[ms] (as the reference), or float64 seconds since
    `orb.epoch` with as_datetime64=False.  Voxels where the solve fails are NaT / NaN."""
    from .utilFcns import lla2ecef
    lon_mesh, lat_mesh, hgt_mesh = np.broadcast_arrays(np.asarray(lon_mesh, float), np.asarray(lat_mesh, float), np.asarray(hgt_mesh, float))
    xyz = np.stack(lla2ecef(lat_mesh, lon_mesh, hgt_mesh), axis=-1)
    _, az, rg = orb.look_vectors(xyz, threshold=1.0e-7, maxiter=100, return_geometry=True, ctx=ctx)
    sec = np.floor((az + rg / SPEED_OF_LIGHT) * 1e3) / 1e3
    if not as_datetime64:
        return sec
    out = np.full(sec.shape, np.datetime64('NaT'), dtype='datetime64[ms]')
    ok = np.isfinite(sec)
    out[ok] = np.datetime64(orb.epoch, 'ms') + np.round(sec[ok] * 1e3).astype('timedelta64[ms]')
    return out


def get_s1_azimuth_time_grid(lon, lat, hgt, datetime, orbit=None, ctx=None):
    """s1_azimuth_timing.py:150-216.  The reference looks the scene's SLC ids and orbit files up on the network; here the
    `Orbit` (or an orbit file path) is an argument.  lon, lat, hgt: all 1-D (-> hgt x lat x lon mesh) or all 3-D."""
    lon, lat, hgt = np.asarray(lon), np.asarray(lat), np.asarray(hgt)
    dims = [c.ndim for c in (lon, lat, hgt)]
    if not all(d == dims[0] for d in dims):
        raise ValueError('All coordinates have same dimension (either 1 or 3 dimensional)')
    if not all(d in (1, 3) for d in dims):
        raise ValueError('Coordinates must be 1d or 3d coordinate arrays')
    if dims[0] == 1:
        hgt, lat, lon = np.meshgrid(hgt, lat, lon, indexing='ij')
    if orbit is None:
        raise NotImplementedError('the SLC / orbit-file lookup (asf_search, s1_orbits) needs the network: pass orbit=')
    if not isinstance(orbit, Orbit):
        orbit = Orbit.from_file(orbit, datetime, pad=600)
    return get_azimuth_time_grid(lon, lat, hgt, orbit, ctx=ctx)


def combine_cubes(cubes, weights, ctx=None):
    """cli/raider.py:817-819 with per-voxel weights: sum(w_i * cube_i) on the device.  cubes: raider_amd.Cube objects on one
    grid; weights: list of (nz, ny, nx) float64 arrays in file order (NumPy or torch on the GPU).  Returns a new f64 Cube."""
    ctx = ctx or cubes[0].ctx
    nd = len(cubes)
    dev = hasattr(weights[0], 'data_ptr')
    if dev:
        import torch
        w = torch.stack([x.to(torch.float64) for x in weights]).contiguous()
        ctx.adopt_torch_stream(w)
    else:
        w = f64(np.stack([np.asarray(x, dtype=np.float64) for x in weights]))
    ny, nx, nz = cubes[0].shape
    if tuple(w.shape) != (nd, nz, ny, nx):
        raise ValueError(f'weights must be {nd} arrays of shape (nz, ny, nx) = ({nz}, {ny}, {nx}); got {tuple(w.shape)}')
    handles = (C.c_void_p * nd)(*[c.handle for c in cubes])
    h = C.c_void_p()
    check(ctx.lib.rdr_cube_blend_weighted(ctx.handle, handles, nd, ptr(w), L.RDR_DEVICE if dev else L.RDR_HOST, C.byref(h)), ctx.handle)
    out = Cube._from_handle(ctx, h)
    out.projection = cubes[0].projection
    return out


def combine_weather_cubes_azimuth_time(pointwise_cubes, total_cubes, times, time_grid, temporal_window_hours=None, ctx=None):
    """The 'azimuth_time_grid' branch of combine_weather_files (cli/raider.py:791-832) without files: `times` = the model
    datetimes of the cubes, `time_grid` = (nz, ny, nx) acquisition times (datetime64, get_azimuth_time_grid).  Returns the
    interpolated (pointwise, total) device cubes - both float64, as in the reference (float64 weight arrays)."""
    if np.any(np.isnat(time_grid)) if np.issubdtype(np.asarray(time_grid).dtype, np.datetime64) else np.any(np.isnan(time_grid)):
        raise ValueError('The Time Grid return nans meaning no orbit was downloaded.')      # cli/raider.py:913-914
    wgts = get_inverse_weights_for_dates(time_grid, list(times), temporal_window_hours=temporal_window_hours, ctx=ctx)
    return combine_cubes(pointwise_cubes, wgts, ctx=ctx), combine_cubes(total_cubes, wgts, ctx=ctx)

"""`getInterpolators` of the delay path (tools/RAiDER/delayFcns.py:23-58).

Returns two interpolator objects with scipy's `RegularGridInterpolator` call semantics (linear,
bounds_error=False, fill_value=nan) and a `.grid` attribute (delay.py:239 reads `interpolators[0].grid[2]`),
both views of ONE device-resident `Cube` (wet and hydro interleaved, gathered together)."""
import os
import threading
from collections import OrderedDict
from pathlib import Path

import numpy as np

from .engine import Cube

# A processed weather-model file is opened by tropo_delay up to three times per call (delay.py:66,76, delayFcns.py:36) and by every
# call of a job that evaluates several AOIs / line-of-sight types on one model epoch.  Both the opened file (its variables, mapped)
# and the device cubes made from it are kept, keyed by the file's identity - resolved path, size, modification time (ns) and inode,
# so a file that was rewritten is read again - for the last RAIDER_HIP_FILE_CACHE files (default 4; 0 switches the cache off).
_CACHE_LOCK = threading.Lock()
_FILE_CACHE = OrderedDict()        # file key -> opened dataset
_CUBE_CACHE = OrderedDict()        # (file key, kind, context serial) -> Cube; a cached cube is shared and NEVER modified (delay._with_model_crs)


def _cache_size():
    try:
        return max(0, int(os.environ.get('RAIDER_HIP_FILE_CACHE', '4')))
    except ValueError:
        return 4


def _file_key(path):
    try:
        st = os.stat(path)
    except OSError:
        return None
    return (os.path.realpath(path), st.st_size, st.st_mtime_ns, st.st_ino)


def _cache_get(cache, key):
    with _CACHE_LOCK:
        if key in cache:
            cache.move_to_end(key)
            return cache[key]
    return None


def _cache_put(cache, key, value, limit):
    with _CACHE_LOCK:
        cache[key] = value
        cache.move_to_end(key)
        while len(cache) > limit:
            cache.popitem(last=False)


def clear_file_cache():
    """Forget every cached weather-model file and the device cubes made from them."""
    with _CACHE_LOCK:
        _FILE_CACHE.clear(); _CUBE_CACHE.clear()


class FieldInterpolator:
    """One field (0 = wet, 1 = hydro) of a device cube, callable like scipy's RGI."""

    def __init__(self, cube, field):
        self.cube = cube
        self.field = field
        self.grid = cube.grid
        self.fill_value = np.nan
        self.bounds_error = False
        self.method = 'linear'

    @property
    def values(self):
        return self.cube.read()[self.field]

    def __call__(self, xi):
        """One gather of THIS field at xi[..., 3] = (y, x, z).  Stateless: the wet and the hydro call of the reference's
        `for intp in interpolators: intp(pts)` (delay.py:213-214,318-319) each upload the points and download their own 8 B per point
        (rdr_interp3 with the other output NULL) - no hand-over between the two objects, so an array edited in place between the calls
        can never be served the other call's result, and nothing is hashed (round 3 hashed all of xi on both calls: 12 ms per
        5 M stations, 80 x the gather).  tropo_delay itself does not come through here: its point branch gathers both fields in one
        launch (Cube.interp_project)."""
        pts = np.asarray(xi, dtype=np.float64)
        return self.cube.interp(pts, field=self.field)[self.field]


class _Var:
    """array + attributes, indexable like a netCDF variable (what tropo_delay reads: var[:] and var.attrs['crs_wkt']).
    `data` may be a zero-argument loader: the array is then read from the file on first use (a ray-traced run never touches
    the f64 `*_total` fields, two thirds of a processed-cube file)."""

    def __init__(self, data, attrs, raw=None):
        self._data, self.attrs, self._raw = data, attrs, raw

    def raw(self):
        """The variable as it lies in the file - a read-only, possibly other-endian view of the file mapping, no copy - or None when
        the file's layout has to be decoded (then `data`).  What the GPU upload of the two big fields takes."""
        return self._raw() if self._raw is not None else None

    @property
    def data(self):
        if callable(self._data):
            self._data = self._data()
        return self._data

    def __getitem__(self, k):
        d = self.data
        if np.ndim(d) == 0 and (k is Ellipsis or k == slice(None)):
            return d                                  # a scalar variable (crs, proj): var[:] / var[...] as netCDF4 / xarray give it
        return d[k]

    def __array__(self, dtype=None, copy=None):
        return np.asarray(self.data, dtype=dtype)


def _read_cube_file(path):
    """Processed weather-model / delay cube file -> {name: _Var}, variables read on first use.  NetCDF-4 (= HDF5, what the
    reference writes, weatherModel.py:659-724) goes through the built-in reader raider_amd.h5lite, NetCDF-3 through scipy."""
    with open(path, 'rb') as fh:
        magic = fh.read(8)
    if magic[:3] == b'CDF':
        from scipy.io import netcdf_file
        f = netcdf_file(str(path), 'r', mmap=True)          # stays open (kept alive by the loaders) until the mapping is dropped

        def loader(name):
            # a private native-endian copy (NetCDF-3 is big-endian): nothing refers to the mapped file afterwards
            return lambda: np.array(f.variables[name].data, dtype=f.variables[name].data.dtype.newbyteorder('='))
        def rawview(name):
            # an OWN read-only mapping of the variable's bytes (big-endian, as NetCDF-3 stores them), independent of scipy's file
            # object: nothing keeps that from closing, and the GPU upload reads the page cache directly
            def get():
                v = f.variables[name]
                d, base = v.data, getattr(f, '_mm_buf', None)
                if d.dtype.kind != 'f' or getattr(v, 'isrec', False) or not isinstance(base, np.ndarray) or not d.flags.c_contiguous:
                    return None
                off = d.__array_interface__['data'][0] - base.__array_interface__['data'][0]
                if off < 0 or off + d.nbytes > base.nbytes:
                    return None
                return np.memmap(str(path), dtype=d.dtype, mode='r', offset=off, shape=d.shape)
            return get
        return {k: _Var(loader(k), {a: (b.decode() if isinstance(b, bytes) else b) for a, b in v._attributes.items()}, rawview(k))
                for k, v in f.variables.items()}
    from . import h5lite
    f = h5lite.File(path)
    out = {}
    for k in f.keys():
        obj = f[k]
        if isinstance(obj, h5lite.Dataset) and obj.dtype is not None and obj.dtype.kind in 'fiu':
            out[k] = _Var(obj.read, {n: v for n, v in obj.attrs.items() if not n.startswith('_N') and n not in ('CLASS', 'NAME')},
                          obj.raw if obj.dtype.kind == 'f' else None)
    return out


def _load_fields(wm_file):
    """Pull x, y, z and the four fields out of a path / xarray.Dataset / mapping."""
    if isinstance(wm_file, (str, Path)):
        limit = _cache_size()
        key = _file_key(wm_file) if limit else None
        ds = _cache_get(_FILE_CACHE, key) if key is not None else None
        if ds is None:
            try:
                import xarray as xr
                ds = xr.load_dataset(wm_file)
            except ImportError:
                ds = _read_cube_file(wm_file)
            if key is not None:
                _cache_put(_FILE_CACHE, key, ds, limit)
    else:
        ds = wm_file
    var = ds.variables if hasattr(ds, 'variables') else ds
    get = lambda k: np.array(var[k][:])
    return var, get


def getInterpolators(wm_file, kind='pointwise', shared=False, ctx=None):
    """delayFcns.py:23-58.  `wm_file`: processed weather-model NetCDF path, an xarray.Dataset, or any mapping
    with x, y, z and wet/hydro (`kind != 'total'`) or wet_total/hydro_total (`kind == 'total'`) in file
    order (z, y, x).  `shared` is accepted and ignored (device memory is shared by construction)."""
    if hasattr(wm_file, 'interpolators') and hasattr(wm_file, 'pointwise'):
        return wm_file.interpolators('total' if kind == 'total' else 'pointwise')    # weather.ProcessedModel: already on the device
    ckey = None
    if isinstance(wm_file, (str, Path)) and _cache_size():
        fkey = _file_key(wm_file)
        if fkey is not None:
            ckey = (fkey, 'total' if kind == 'total' else 'pointwise', ctx.serial if ctx is not None else None)      # (a serial, not id(): never recycled)
            cube = _cache_get(_CUBE_CACHE, ckey)
            if cube is not None:                                       # the same file, still on the device
                if cube.has_nan():
                    from .logger import logger
                    logger.critical('Weather model contains NaNs!')
                return FieldInterpolator(cube, 0), FieldInterpolator(cube, 1)
    var, get = _load_fields(wm_file)
    xs, ys, zs = get('x'), get('y'), get('z')

    def field(name):
        # the two big fields straight from the file mapping when the file allows it (contiguous data, any byte order): uploaded as
        # they lie there, byte-swapped and NaN-scanned on the device - no host pass over 58 MB (ERA5-sized) to 400 MB (HRRR-sized)
        v = var[name]
        r = v.raw() if hasattr(v, 'raw') else None
        return r if r is not None else get(name)
    wet = field('wet_total' if kind == 'total' else 'wet')
    hydro = field('hydro_total' if kind == 'total' else 'hydro')
    cube = Cube(ys, xs, zs, wet, hydro, order='zyx', ctx=ctx)      # no host transpose (delayFcns.py:40-41 does one)
    if cube.has_nan():                                             # delayFcns.py:50-52, answered by the packing kernel
        from .logger import logger
        logger.critical('Weather model contains NaNs!')
    if ckey is not None:
        _cache_put(_CUBE_CACHE, ckey, cube, 2 * _cache_size())     # (two kinds per file)
    return FieldInterpolator(cube, 0), FieldInterpolator(cube, 1)


def interpolators_from_cube(cube):
    """Wrap an existing device `Cube` (e.g. a blended one) as the (ifWet, ifHydro) pair."""
    return FieldInterpolator(cube, 0), FieldInterpolator(cube, 1)

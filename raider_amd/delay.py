"""RAiDER tropospheric delay calculation on MI355X - the drop-in for tools/RAiDER/delay.py.

Same functions, signatures and quirks as the reference module:
  tropo_delay (delay.py:35-130), _get_delays_on_cube (:133-193), _build_cube (:196-216),
  _build_cube_ray (:219-326), writeResultsToXarray (:329-401), transformPoints (:404-436);
`getDelays` is kept as a legacy alias of `tropo_delay`.
Host code only orchestrates; every gather / ray march happens in the HIP kernels (raider_amd/csrc).
"""
import datetime as dt
import os

import numpy as np

from . import _pinned
from ._lib import FLAG_DIVERGED, NoLevels
from .constants import _ZREF
from .delayFcns import FieldInterpolator, getInterpolators, _load_fields
from .engine import Cube, Rays
from .logger import logger
from .utilFcns import ecef2lla, lla2ecef

try:  # optional: only needed for CRSs other than EPSG:4326 / EPSG:4978
    import pyproj
except ImportError:  # pragma: no cover - pyproj is absent from the build image
    pyproj = None


# ------------------------------------------------------------------------------------------------
# CRS helpers (the reference uses pyproj.CRS objects; EPSG ints / 'EPSG:xxxx' strings are enough here)
# ------------------------------------------------------------------------------------------------
def _epsg(crs):
    """EPSG code of a CRS-like (int, 'EPSG:4326', '4326', pyproj.CRS, object with to_epsg) or None."""
    if crs is None:
        return None
    if isinstance(crs, (int, np.integer)):
        return int(crs)
    if isinstance(crs, str):
        s = crs.split(':')[-1]
        try:
            return int(s)
        except ValueError:
            if pyproj is not None:
                return pyproj.CRS(crs).to_epsg()
            return None
    if hasattr(crs, 'to_epsg'):
        return crs.to_epsg()
    return None


def _same_crs(a, b):
    ea, eb = _epsg(a), _epsg(b)
    if ea is not None and eb is not None:
        return ea == eb
    return a == b


def _is_4326(crs):
    return _epsg(crs) == 4326


def _crs_dict(crs):
    """PROJ parameters of a CRS given as a dict, a PROJ string ('+proj=lcc +lat_1=... +a=... +b=...', models/hrrr.py:255-259,
    '+proj=stere +lat_0=90 ...', models/hrrr.py:22-25) or a pyproj CRS; None when it cannot be taken apart."""
    if isinstance(crs, dict):
        return dict(crs)
    if isinstance(crs, str) and '+proj=' in crs:
        d = {}
        for tok in crs.split():
            if tok.startswith('+') and '=' in tok:
                k, v = tok[1:].split('=', 1)
                try:
                    d[k] = float(v)
                except ValueError:
                    d[k] = v
            elif tok.startswith('+') and len(tok) > 1:
                d[tok[1:]] = True                                  # flags: +south, +no_defs
        return d
    if pyproj is not None and hasattr(crs, 'to_dict'):
        try:
            return crs.to_dict()
        except Exception:
            return None
    return None


def _ellipsoid(d):
    a = float(d.get('a', d.get('R', 6378137.0)))
    if 'b' in d:
        b = float(d['b']); es = 1.0 - (b * b) / (a * a)
    elif 'rf' in d:
        f = 1.0 / float(d['rf']); es = 2 * f - f * f
    elif 'es' in d:
        es = float(d['es'])
    elif 'R' in d or 'a' in d:
        es = 0.0 if 'R' in d or 'ellps' not in d else 0.0066943799901413165
    else:
        es = 0.0066943799901413165     # WGS84 default ellipsoid
    return a, max(es, 0.0)


def _lcc_params(crs):
    """Lambert-conformal-conic parameters of a model CRS, or None."""
    d = _crs_dict(crs)
    if not d or d.get('proj') != 'lcc':
        return None
    a, es = _ellipsoid(d)
    a *= float(d.get('k_0', d.get('k', 1.0)))          # 1SP form: the scale factor multiplies every radius (rho = k_0 a F t^n), nothing else
    lat_1 = float(d.get('lat_1', d.get('lat_0', 0.0)))
    return dict(lat_1=lat_1, lat_2=float(d.get('lat_2', lat_1)), lat_0=float(d.get('lat_0', 0.0)), lon_0=float(d.get('lon_0', 0.0)),
                x_0=float(d.get('x_0', 0.0)), y_0=float(d.get('y_0', 0.0)), a=a, es=es)


def _stere_params(crs):
    """Polar-stereographic parameters of a model CRS (HRRR-AK, models/hrrr.py:22-25), or None.  The oblique aspect raises."""
    d = _crs_dict(crs)
    if not d or d.get('proj') not in ('stere', 'ups'):
        return None
    a, es = _ellipsoid(d)
    lat_0 = float(d.get('lat_0', 0.0))
    if abs(abs(lat_0) - 90.0) > 1e-9:
        raise NotImplementedError(f'only the polar aspect of the stereographic projection is built in (lat_0 = +-90), got lat_0 = {lat_0}')
    lat_ts = d.get('lat_ts')
    return dict(lat_0=lat_0, lat_ts=None if lat_ts is None else float(lat_ts), k_0=float(d.get('k_0', d.get('k', 1.0))),
                lon_0=float(d.get('lon_0', 0.0)), x_0=float(d.get('x_0', 0.0)), y_0=float(d.get('y_0', 0.0)), a=a, es=es)


def _tm_params(crs):
    """Transverse-Mercator parameters of a CRS: a WGS 84 / UTM EPSG code, or a `+proj=tmerc` / `+proj=utm` PROJ string / dict; else None."""
    from .utilFcns import utm_params
    e = _epsg(crs)
    if e is not None:
        return utm_params(e)
    d = _crs_dict(crs)
    if not d or d.get('proj') not in ('tmerc', 'etmerc', 'utm'):
        return None
    a, es = _ellipsoid(d)
    if d.get('proj') == 'utm':
        zone = int(d['zone'])
        return dict(a=a, es=es, lat_0=0.0, lon_0=6.0 * zone - 183.0, k_0=0.9996, x_0=500000.0, y_0=10000000.0 if ('south' in d) else 0.0)
    return dict(a=a, es=es, lat_0=float(d.get('lat_0', 0.0)), lon_0=float(d.get('lon_0', 0.0)), k_0=float(d.get('k_0', d.get('k', 1.0))),
                x_0=float(d.get('x_0', 0.0)), y_0=float(d.get('y_0', 0.0)))


def _model_projection(model_crs):
    """The device-side description of a model CRS: None for lon/lat, a dict(proj='lcc' | 'stere', ...) for the conic grids the kernels
    project to, False for anything else."""
    if _is_4326(model_crs):
        return None
    lcc = _lcc_params(model_crs)
    if lcc is not None:
        return dict(lcc, proj='lcc')
    st = _stere_params(model_crs)
    if st is not None:
        return dict(st, proj='stere')
    return False


def _with_model_crs(cube, model_crs):
    """(the cube THIS call works on, model is projected).  The reference rebuilds its transformers in every call and shares nothing
    (delay.py:196-216,238-253); here the device cube of a file is cached and shared between calls and threads, so it is NEVER
    modified: a call whose model CRS differs from what the cube carries gets a view - a second handle on the same device buffers
    with its own projection (Cube.view, rdr_cube_view; a few microseconds, no device work).  An unknown CRS: (cube, False)."""
    want = _model_projection(model_crs)
    if want is False:
        return cube, False
    have = cube.projection
    same = (have is None and want is None) or (have is not None and want is not None and have.get('proj') == want['proj'] and
                                               all(have.get(k) == v for k, v in want.items()))
    return (cube if same else cube.view(want)), want is not None


# ------------------------------------------------------------------------------------------------
# small containers (the AOI / Dataset providers themselves are outside the hot path)
# ------------------------------------------------------------------------------------------------
class GridAOI:
    """Array-backed stand-in for llreader.BoundingBox / Geocube: output grid nodes only."""

    def __init__(self, xpts, ypts, heights=None):
        self.xpts = np.asarray(xpts, dtype=np.float64)
        self.ypts = np.asarray(ypts, dtype=np.float64)
        self._heights = heights

    def type(self):
        return 'bounding_box' if self._heights is None else 'Geocube'

    def readZ(self):
        return self._heights


class PointsAOI:
    """Array-backed stand-in for llreader.StationFile / RasterRDR: query points + the intermediate grid."""

    def __init__(self, lats, lons, hgts, xpts=None, ypts=None):
        self._lats, self._lons, self._hgts = (np.asarray(a, dtype=np.float64) for a in (lats, lons, hgts))
        if xpts is not None:
            self.xpts = np.asarray(xpts, dtype=np.float64)
            self.ypts = np.asarray(ypts, dtype=np.float64)

    def type(self):
        return 'station_file'

    def readLL(self):
        return self._lats, self._lons

    def readZ(self):
        return self._hgts

    # llreader.py:173-191 (EPSG:4326 only): grid from the bounding box of the points
    def set_output_spacing(self, ll_res=None):
        self._spacing = ll_res

    def set_output_xygrid(self, dst_crs=4326):
        def lohi(a):           # np.nanmin / np.nanmax (llreader.py:173-191) - the plain reductions first: 4x cheaper, and right unless a NaN is there
            lo, hi = a.min(), a.max()
            return (lo, hi) if (lo == lo and hi == hi) else (np.nanmin(a), np.nanmax(a))
        (S, N), (W, E) = lohi(self._lats), lohi(self._lons)
        sp = self._spacing
        self.xpts = np.arange(W, E + sp, sp)
        self.ypts = np.arange(N, S - sp, -sp)


class DelayCube:
    """What writeResultsToXarray returns when xarray is not installed: the same variables/coords,
    readable by getInterpolators (`.variables[name][:]`)."""

    def __init__(self, variables, attrs):
        self.variables = variables
        self.attrs = attrs

    def __getitem__(self, k):
        return self.variables[k]

    def __getattr__(self, k):
        try:
            return self.__dict__['variables'][k]
        except KeyError:
            raise AttributeError(k)

    def to_netcdf(self, path, format='NETCDF4'):
        """The delay-cube file of delay.py:329-401 / cli/raider.py:373-398 (`ds.to_netcdf`): dims z, y, x; variables wet, hydro (f64,
        units m, grid_mapping crs), coordinate variables, the integer `crs` grid-mapping variable and the global attributes.
        format='NETCDF4' (what the reference writes): HDF5 through raider_amd.h5write; 'NETCDF3_64BIT': classic format through
        scipy.  (When xarray is installed writeResultsToXarray returns a real Dataset instead of this class.)"""
        v = self.variables
        degrees = self.attrs.get('_degrees', True)
        desc = str(self.attrs.get('description', '')).replace('RAiDER geo cube - ', '')
        gattrs = {k: str(val) for k, val in self.attrs.items() if not k.startswith('_')}   # (_degrees / _crs_cf steer the writer)
        coord_attrs = {'z': dict(axis='Z', units='m', description='height above ellipsoid'),
                       'y': dict(units='degrees_north', standard_name='latitude', long_name='latitude') if degrees else
                            dict(axis='Y', standard_name='projection_y_coordinate', long_name='y-coordinate in projected coordinate system', units='m'),
                       'x': dict(units='degrees_east', standard_name='longitude', long_name='longitude') if degrees else
                            dict(axis='X', standard_name='projection_x_coordinate', long_name='x-coordinate in projected coordinate system', units='m')}
        if format.upper().startswith('NETCDF4'):
            from .h5write import write_netcdf4
            variables = {d: ((d,), np.asarray(v[d], dtype=np.float64), coord_attrs[d]) for d in ('z', 'y', 'x')}
            for name, long in (('wet', 'wet'), ('hydro', 'hydrostatic')):
                variables[name] = (('z', 'y', 'x'), np.asarray(v[name], dtype=np.float64),
                                   dict(units='m', description=f'{long} {desc} delay', grid_mapping='crs'))
            variables['crs'] = ((), np.array(-2147483647, dtype=np.int64), dict(self.attrs.get('_crs_cf') or {}))
            write_netcdf4(path, {d: int(np.size(v[d])) for d in ('z', 'y', 'x')}, variables, gattrs)
            return str(path)
        from scipy.io import netcdf_file
        with netcdf_file(str(path), 'w', version=2) as f:
            for k, val in gattrs.items():
                setattr(f, k, val)
            for d in ('z', 'y', 'x'):
                f.createDimension(d, int(np.size(v[d])))
                cv = f.createVariable(d, 'f8', (d,))
                cv[:] = np.asarray(v[d], dtype=np.float64)
                for k, val in coord_attrs[d].items():
                    setattr(cv, k, val)
            for name, long in (('wet', 'wet'), ('hydro', 'hydrostatic')):
                dv = f.createVariable(name, 'f8', ('z', 'y', 'x'))
                dv[:] = np.asarray(v[name], dtype=np.float64)
                dv.units = 'm'; dv.description = f'{long} {desc} delay'; dv.grid_mapping = 'crs'
            crs = f.createVariable('crs', 'i4', ())
            crs.data[()] = -2147483647
            for k, val in (self.attrs.get('_crs_cf') or {}).items():
                setattr(crs, k, np.float64(val) if isinstance(val, float) else val)      # (scipy narrows a bare Python float to float32)
        return str(path)


def _is_cube_aoi(aoi):
    """delay.py:98 `isinstance(aoi, (BoundingBox, Geocube))`, duck-typed on AOI.type() (llreader.py:50-51,316,373)."""
    t = getattr(aoi, 'type', None)
    t = t() if callable(t) else t
    return t in ('bounding_box', 'Geocube')


def _is_geocube(aoi):
    t = getattr(aoi, 'type', None)
    t = t() if callable(t) else t
    return t == 'Geocube'


# ------------------------------------------------------------------------------------------------
# interpolator plumbing
# ------------------------------------------------------------------------------------------------
def _cube_of(interpolators):
    """(Cube, field index per interpolator).  Accepts this package's FieldInterpolators, or ANY objects with
    scipy's `.grid` / `.values` (e.g. the scipy RGIs the reference builds): their data is uploaded once and
    cached on the first object - the arithmetic still runs on the GPU, never in scipy."""
    first = interpolators[0]
    if all(isinstance(i, FieldInterpolator) for i in interpolators) and all(i.cube is first.cube for i in interpolators):
        return first.cube, [i.field for i in interpolators]
    if len(interpolators) > 2:
        raise ValueError('at most two interpolators (wet, hydro) are supported')
    # Foreign interpolators are uploaded afresh on EVERY call: the reference reads `.values` at call time, so an array edited in place
    # between two calls must give new results, and no identity / shape / sample check can see such an edit short of reading all of
    # it - which costs what the upload costs (57 MB of an ERA5-sized pair: ~2 ms; the device buffer is recycled from the context's
    # cube pool, nothing is allocated).  (Rounds 3-4 cached the upload on the first object, validated by identity of `.values` only.)
    last = interpolators[-1]
    grid = first.grid
    a = np.asarray(first.values)
    b = np.asarray(last.values)
    cube = Cube(grid[0], grid[1], grid[2], a, b.astype(a.dtype, copy=False), order='yxz')
    return cube, list(range(len(interpolators)))


# ------------------------------------------------------------------------------------------------
# public API
# ------------------------------------------------------------------------------------------------
def tropo_delay(datetime, weather_model_file, aoi, los, height_levels=None, out_proj=4326, zref=None):
    """delay.py:35-130: ZTD, projected STD, or ray-traced STD on an AOI.

    weather_model_file: path to a processed weather-model NetCDF, an xarray.Dataset, or a mapping with
    x, y, z, wet, hydro, wet_total, hydro_total (file order (z,y,x)) and optionally a 'proj' entry.
    Returns (Dataset-like, None) for cube AOIs, else (wetDelay, hydroDelay) at the query points."""
    crs = out_proj
    var, get = _load_fields(weather_model_file)
    # CRS of the weather model (delay.py:66-73)
    wm_proj = None
    try:
        pj = var['proj']
        if isinstance(pj, (str, dict, int)):                       # mapping input: CRS given directly (EPSG / PROJ string / dict)
            wm_proj = pj
        else:
            from .crs import crs_from_proj_var
            wm_proj = crs_from_proj_var(pj.attrs)                  # pyproj.CRS.from_wkt when installed; else CF attributes / WKT reader
    except (KeyError, AttributeError, TypeError):
        logger.warning("WARNING: I can't find a CRS in the weather model file, so I will assume you are using WGS84")
        wm_proj = 4326

    wm_levels = get('z')
    toa = wm_levels.max() - 1                                   # delay.py:78
    if height_levels is None:
        height_levels = aoi.readZ() if _is_geocube(aoi) else wm_levels
    if zref is None:
        zref = toa
    if zref > toa:
        zref = toa
        logger.warning(f'Requested integration height (zref) is higher than top of weather model. Forcing to top ({toa}).')

    if _is_cube_aoi(aoi):
        ds = _get_delays_on_cube(datetime, weather_model_file, wm_proj, aoi, height_levels, los, crs, zref, _loaded=var)
        return ds, None

    # point branch (delay.py:101-128): the intermediate cube, interpolated to the query points.  The cube never leaves the device:
    # zenith / projected lines of sight run the whole branch in ONE library call (rdr_point_delays: _build_cube into device scratch,
    # one gather of both fields at the points, the division by cos(inc) in the same launch, the points travelling up while the cube
    # is built); ray-traced ones keep the cube as a device `Cube` (rdr_raytrace_slices_to_cube) and gather from it.  The points go
    # up, 2 x N doubles come down.  (Round 3 mirrored the reference's data flow literally: cube down, Dataset, cube up, two gathers.)
    lats, lons = aoi.readLL()
    hgts = aoi.readZ()
    proj = None
    if los.is_Projected():
        los.setTime(datetime)
        los.setPoints(lats, lons, hgts)
        proj = los._divisor_source() if hasattr(los, '_divisor_source') else False
    kw = {} if not proj else ({'inc': proj[1]} if proj[0] == 'inc' else {'divisor': proj[1]})
    # transformPoints(4326 -> 4326) is the identity stack: the three arrays go up as they are
    pts = (lats, lons, hgts) if _is_4326(out_proj) else (transformPoints(lats, lons, hgts, 4326, out_proj),)
    res = _point_branch_on_device(weather_model_file, wm_proj, aoi, height_levels, los, crs, zref, var, pts, kw)
    if res is None:
        # jobs the device route does not take (an output CRS that is neither the model's nor lon/lat, a one-node grid axis, > 512
        # heights): the reference's own sequence
        ds = _get_delays_on_cube(datetime, weather_model_file, wm_proj, aoi, height_levels, los, crs, zref, _loaded=var)
        try:
            ifWet, ifHydro = getInterpolators(ds, 'ztd')
        except RuntimeError:
            raise RuntimeError(f'Failed to get weather model {weather_model_file} interpolators.')
        res = ifWet.cube.interp_project(*pts, **kw)
    wetDelay, hydroDelay = res
    if proj is False:                                  # a foreign projected LOS object: its own __call__ (losreader.py:110-133)
        wetDelay = los(wetDelay)
        hydroDelay = los(hydroDelay)
    return wetDelay, hydroDelay


getDelays = tropo_delay   # legacy name used by BASELINE.json's north_star


class _Result(list):
    """[wetDelay, hydroDelay] as the reference's _build_cube / _build_cube_ray return it, plus what the device already knows about it:
    `has_nan` = np.isnan(...).any() over both arrays (delay.py:187), scanned before the download; None: unknown (the host scans)."""
    has_nan = None


def _has_nan(a):
    """np.isnan(a).any() (delay.py:187) without the boolean temporary: a NaN anywhere makes the (threaded BLAS) dot product of
    the array with itself NaN, and nothing else does (squares cannot cancel); non-f64 / tiny inputs take the plain route."""
    a = np.asarray(a)
    if a.dtype != np.float64 or a.size < (1 << 16) or not a.flags.c_contiguous:
        return bool(np.isnan(a).any())
    flat = a.reshape(-1)
    return bool(np.isnan(np.dot(flat, flat)))


def _ensure_output_grid(aoi, weather_model_file, crs):
    """delay.py:142-151: an AOI without an output grid gets one at the weather model's own spacing."""
    try:
        aoi.xpts
    except AttributeError:
        _, get = _load_fields(weather_model_file)
        x_spacing = np.diff(get('x')).mean()
        y_spacing = np.diff(get('y')).mean()
        aoi.set_output_spacing(ll_res=np.min([x_spacing, y_spacing]))
        aoi.set_output_xygrid(crs)


def _raise_slice_failures(K, flags, zz, top):
    """The reference's failure modes of one ray-traced slice batch (delay.py:276-283), in slice order."""
    from ._lib import FLAG_ANY_FINITE, FLAG_ANY_NAN
    for hh, ht in enumerate(zz):
        if K[hh] == 0:
            if ht == top:                                              # delay.py:276-277: the slice stays zero (the kernels wrote 0)
                continue
            raise TypeError("ufunc 'isnan' not supported for the input types (build_ray returned None)")   # delay.py:279
        if not (flags[hh] & FLAG_ANY_FINITE):
            raise ValueError('geo2rdr did not converge. Check orbit coverage')            # delay.py:279-280
        if flags[hh] & FLAG_ANY_NAN:
            raise ValueError('some ray lengths are NaN: the number of integration parts (delay.py:283) is undefined')
        if flags[hh] & FLAG_DIVERGED:
            raise ValueError('ray lengths diverged: a model level asks for fewer than 2 or more than 65536 integration parts '
                             '(are the look vectors unit vectors?)')


def _point_branch_on_device(weather_model_file, wm_proj, aoi, heights, los, crs, zref, _loaded, pts, kw):
    """_get_delays_on_cube (delay.py:133-193) + the second-stage interpolation (delay.py:110-128) with the intermediate cube - what
    getInterpolators(ds, 'ztd') would wrap, axes (aoi.ypts, aoi.xpts, heights) - left ON THE DEVICE.  `pts`: (y, x, z) arrays or one
    packed array, in the output CRS; `kw`: inc= / divisor= of a projected line of sight.  Returns (wetDelay, hydroDelay), or None for
    the jobs that need the host sequence (an output grid that is neither the model's CRS nor lon/lat, a one-node axis, > 512
    heights).  Same kernels, same arithmetic as _build_cube / _build_cube_ray + the interpolators: the values are theirs bit for bit."""
    zpts = np.array(heights, dtype=np.float64)
    if _loaded is not None and not isinstance(weather_model_file, (str, os.PathLike)):
        weather_model_file = _loaded
    _ensure_output_grid(aoi, weather_model_file, crs)
    xpts, ypts = np.asarray(aoi.xpts, dtype=np.float64), np.asarray(aoi.ypts, dtype=np.float64)
    if zpts.ndim != 1 or min(xpts.size, ypts.size, zpts.size) < 2 or zpts.size > 512 or xpts.size + ypts.size + zpts.size > 100000:
        return None
    dz = np.diff(zpts)
    if not (np.all(dz > 0) or np.all(dz < 0)):
        return None                                                    # (scipy's grid rule: the host sequence raises what it raises)
    from ._lib import DeviceOutOfMemory
    try:
        if los.is_Zenith() or los.is_Projected():
            ifWet, ifHydro = getInterpolators(weather_model_file, 'total')
            cube = ifWet.cube
            if _same_crs(wm_proj, crs) and (cube.projection is None or _is_4326(wm_proj)):
                cube, _ = _with_model_crs(cube, 4326)                     # grid nodes already in the model's coordinates: nothing to project
            elif _is_4326(crs):
                cube, projected = _with_model_crs(cube, wm_proj)
                if not projected:
                    return None
            else:
                return None
            wet, hyd, has_nan = cube.point_delays(xpts, ypts, zpts, *pts, **kw)
        else:
            if not (_is_4326(crs) and hasattr(los, 'ray_batch_slices')):
                return None
            ifWet, ifHydro = getInterpolators(weather_model_file, kind='pointwise')
            cube, projected = _with_model_crs(ifWet.cube, wm_proj)
            if not (projected or _is_4326(wm_proj)):
                return None
            # the whole intermediate cube in one batch only when it fits the slice budget of _build_cube_ray (64 B per ray and slice on
            # the device next to the 16 B per cell of the cube itself); larger jobs take the chunked host sequence
            if xpts.size * ypts.size * zpts.size * 80 > int(os.environ.get('RAIDER_HIP_SLICE_BUDGET_BYTES', 8 << 30)):
                return None
            rays = los.ray_batch_slices(xpts, ypts, zpts)
            dcube, K, _nparts, flags = cube.raytrace_slices_to_cube(rays, zpts, zref, 1000.0)
            _raise_slice_failures(K, flags, zpts, zpts[-1])
            wet, hyd = dcube.interp_project(*pts, **kw)
            has_nan = dcube.has_nan()                                      # (asked AFTER the gather: the cube was made without a host synchronisation)
    except (MemoryError, RuntimeError) as exc:
        # out of DEVICE memory (RDR_ERR_OOM -> _lib.DeviceOutOfMemory, a MemoryError; torch's allocator: torch.OutOfMemoryError, a
        # RuntimeError): the one-call route holds the whole intermediate cube; the host sequence builds it in chunks that are halved
        # until they fit (_build_cube_ray), so the job still completes - as it did before the cube stayed on the device
        if not (isinstance(exc, DeviceOutOfMemory) or type(exc).__name__ == 'OutOfMemoryError'):
            raise
        logger.info(f'the point branch did not fit the device in one piece ({exc}); continuing with the chunked sequence')
        try:
            cube.ctx.trim(0)
        except Exception:
            pass
        return None
    if has_nan:                                                        # delay.py:187, answered while the cube was packed
        logger.critical('There are missing delay values. Check your inputs.')
    return wet, hyd


def _get_delays_on_cube(datetime, weather_model_file, wm_proj, aoi, heights, los, crs, zref, nproc=1, _loaded=None):
    """delay.py:133-193.  `_loaded`: the already-opened variables of `weather_model_file` (tropo_delay opens the file once;
    the reference loads it three times, delay.py:66,76 and delayFcns.py:36)."""
    zpts = np.array(heights)
    wm_source = weather_model_file
    if _loaded is not None and not isinstance(weather_model_file, (str, os.PathLike)):
        weather_model_file = _loaded          # (a path goes on as a path: the opened file and its device cubes are cached by file identity)
    _ensure_output_grid(aoi, weather_model_file, crs)

    if los.is_Zenith() or los.is_Projected():
        # NB: a projected LOS on a cube AOI yields ZENITH delays, exactly like the reference
        out_type = 'zenith' if los.is_Zenith() else 'slant - projected'
        ifWet, ifHydro = getInterpolators(weather_model_file, 'total')
        res = _build_cube(aoi.xpts, aoi.ypts, zpts, wm_proj, crs, [ifWet, ifHydro])
    else:
        out_type = 'slant - raytracing'
        ifWet, ifHydro = getInterpolators(weather_model_file, kind='pointwise', shared=(nproc > 1))
        if nproc == 1:
            res = _build_cube_ray(aoi.xpts, aoi.ypts, zpts, los, wm_proj, crs, [ifWet, ifHydro], MAX_TROPO_HEIGHT=zref)
        else:
            raise NotImplementedError     # delay.py:178-185 (multi-GPU: see raider_amd.distributed)
    wetDelay, hydroDelay = res

    known = getattr(res, 'has_nan', None)           # the device's scan of exactly these arrays, carried by the result itself
    if known if known is not None else (_has_nan(wetDelay) or _has_nan(hydroDelay)):
        logger.critical('There are missing delay values. Check your inputs.')

    return writeResultsToXarray(datetime, aoi.xpts, aoi.ypts, zpts, crs, wetDelay, hydroDelay, wm_source, out_type)


def _build_cube(xpts, ypts, zpts, model_crs, pts_crs, interpolators):
    """delay.py:196-216: zenith / projected cube, one trilinear gather of both fields per node."""
    cube, fields = _cube_of(interpolators)
    zpts = np.asarray(zpts)
    if _is_4326(model_crs):
        cube, _ = _with_model_crs(cube, 4326)              # (a cube that carries a projection: this call gets an unprojected view of it)
    def hinted(c):
        res = c.build_cube(xpts, ypts, zpts, want_nan=True)
        out = _Result(res[f] for f in fields)
        if len(out) == 2:                                 # what np.isnan(result).any() would find (delay.py:187): known from the device scan
            out.has_nan = res[2]                          # (part of the call's own result: the cube may be serving other threads)
        return out
    if _same_crs(model_crs, pts_crs) and cube.projection is None:
        return hinted(cube)                                # points generated on the fly in the kernel
    if _is_4326(pts_crs):
        pcube, projected = _with_model_crs(cube, model_crs)
        if projected:
            return hinted(pcube)                           # lon/lat nodes projected to the model's LCC grid on the device
    xx, yy = np.meshgrid(xpts, ypts)
    outputArrs = [np.zeros((zpts.size, len(ypts), len(xpts))) for _ in interpolators]
    for ii, ht in enumerate(zpts):
        pts = transformPoints(yy, xx, np.full(yy.shape, ht), pts_crs, model_crs)    # delay.py:207-209
        res = cube.interp(pts)
        for mm, f in enumerate(fields):
            outputArrs[mm][ii, ...] = res[f]
    return outputArrs


def _build_cube_ray(xpts, ypts, zpts, los, model_crs, pts_crs, interpolators, outputArrs=None,
                    MAX_SEGMENT_LENGTH=1000.0, MAX_TROPO_HEIGHT=_ZREF):
    """delay.py:219-326: ray-traced cube.  One fused GPU pass pair per height slice:
    pass 1 = build_ray's per-level ray lengths reduced to the slice maximum (-> nParts, delay.py:283),
    pass 2 = Newton level intersections + ECEF->geodetic + trilinear gather + trapezoid, per ray."""
    cube, fields = _cube_of(interpolators)
    cube, projected = _with_model_crs(cube, model_crs)     # (never modifies a shared cube: a view with THIS call's model CRS)
    if not (projected or _is_4326(model_crs)):
        raise NotImplementedError('ray tracing needs the weather cube on an EPSG:4326 lat/lon grid, a Lambert-conformal-conic grid '
                                  f'(HRRR) or a polar-stereographic grid (HRRR-AK); got {model_crs!r}')
    xpts = np.asarray(xpts, dtype=np.float64)
    ypts = np.asarray(ypts, dtype=np.float64)
    zpts = np.asarray(zpts)
    output_created_here = False
    if outputArrs is None:
        output_created_here = True
        # the reference starts from zeros and accumulates (delay.py:245-248,323); a fresh cube is simply written slice by slice
        # (large cubes in recycled page-locked memory: no first-touch page faults, downloads overlap the kernels - _pinned.py)
        outputArrs = [_pinned.empty((zpts.size, ypts.size, xpts.size)) for _ in interpolators]
    direct = output_created_here and list(fields) == [0, 1]

    grid_is_ll = _is_4326(pts_crs)
    if direct and grid_is_ll and hasattr(los, 'ray_batch_slices') and zpts.size > 0:
        # the whole height loop as one batched launch pair per <= 512 slices (Cube.raytrace_slices): bit-identical to the slice
        # loop below, but a production job (20 heights x 1e4-1e5 rays) fills the GPU instead of a tenth of it
        any_nan = False
        # slices per call from a byte budget, not a fixed count: a batch holds per ray and slice its look vector (orbit-based ones
        # depend on the height: 24 B, plus the 24 B target they were solved from) and 16 B of delays on the device, next to the 232 B
        # ray records the library chunks by itself.  RAIDER_HIP_SLICE_BUDGET_BYTES (default 8 GiB); a batch the device still cannot
        # hold is halved until it fits.
        n_per = max(1, xpts.size * ypts.size)
        budget = int(os.environ.get('RAIDER_HIP_SLICE_BUDGET_BYTES', 8 << 30))
        chunk = int(min(512, max(1, budget // (n_per * 64))))
        s0 = 0
        while s0 < zpts.size:
            zz = np.ascontiguousarray(zpts[s0:s0 + chunk], dtype=np.float64)
            logger.info(f'Processing slices {s0 + 1}-{s0 + zz.size} / {len(zpts)}')
            try:
                rays = los.ray_batch_slices(xpts, ypts, zz)
                if rays._torch_device is None:
                    _, _, K, _nparts, flags, nan_out = cube.raytrace_slices(rays, zz, MAX_TROPO_HEIGHT, MAX_SEGMENT_LENGTH, want_nan=True,
                                                                            out=(outputArrs[0][s0:s0 + zz.size], outputArrs[1][s0:s0 + zz.size]))
                    any_nan = any_nan or bool(nan_out.any())
                else:           # a device-resident batch (orbit-based look vectors made on the GPU): device outputs, one download per field
                    import torch
                    dw, dh, K, _nparts, flags, nan_out = cube.raytrace_slices(rays, zz, MAX_TROPO_HEIGHT, MAX_SEGMENT_LENGTH, want_nan=True)
                    any_nan = any_nan or bool(nan_out.any())
                    torch.from_numpy(outputArrs[0][s0:s0 + zz.size]).copy_(dw); torch.from_numpy(outputArrs[1][s0:s0 + zz.size]).copy_(dh)
                    del dw, dh
            except (MemoryError, RuntimeError) as exc:
                # out of DEVICE memory, told by status / class, not by the wording of a message: RDR_ERR_OOM arrives as
                # _lib.DeviceOutOfMemory (a MemoryError), torch's allocator raises torch.OutOfMemoryError (a RuntimeError)
                if chunk == 1 or not (isinstance(exc, MemoryError) or type(exc).__name__ == 'OutOfMemoryError'):
                    raise
                chunk = max(1, chunk // 2)                                 # the device could not hold the batch: smaller ones
                logger.info(f'slice batch did not fit the device ({exc}); continuing with {chunk} slices per call')
                continue
            s0 += zz.size
            _raise_slice_failures(K, flags, zz, zpts[-1])
        # what np.isnan(result).any() would find (delay.py:187), already known from the device-side scan of every slice
        outputArrs = _Result(outputArrs)
        outputArrs.has_nan = any_nan
        return outputArrs
    for hh, ht in enumerate(zpts):
        logger.info(f'Processing slice {hh + 1} / {len(zpts)}: {ht}')
        if grid_is_ll and hasattr(los, 'ray_batch'):
            rays = los.ray_batch(xpts, ypts, ht)                           # look vectors made / read on the device
        else:
            xx, yy = np.meshgrid(xpts, ypts)
            if grid_is_ll:
                llh = [xx, yy, np.full(yy.shape, ht)]
            else:                                                          # delay.py:262-263
                p = transformPoints(yy, xx, np.full(yy.shape, ht), pts_crs, 4326)
                llh = [p[..., 1], p[..., 0], p[..., 2]]
            xyz = np.stack(lla2ecef(llh[1], llh[0], llh[2]), axis=-1)      # delay.py:267
            LOS = los.getLookVectors(ht, llh, xyz, yy)                     # delay.py:270
            if grid_is_ll:
                rays = Rays.grid(xpts, ypts, los=LOS)
            else:
                rays = Rays.points(lat=llh[1], lon=llh[0], los=LOS)
        on_device = getattr(rays, '_torch_device', None) is not None
        try:
            wet, hyd, _nparts, _flags = cube.raytrace(rays, ht, MAX_TROPO_HEIGHT, MAX_SEGMENT_LENGTH,
                                                      out=(outputArrs[0][hh], outputArrs[1][hh]) if (direct and not on_device) else None)
        except NoLevels:
            if ht == zpts[-1]:                                             # delay.py:276-277: the slice stays zero
                if output_created_here:
                    for arr in outputArrs:
                        arr[hh, ...] = 0.0
                continue
            raise TypeError("ufunc 'isnan' not supported for the input types (build_ray returned None)")   # delay.py:279
        if on_device:
            wet, hyd = wet.cpu().numpy(), hyd.cpu().numpy()
            if direct:
                outputArrs[0][hh, ...] = wet; outputArrs[1][hh, ...] = hyd
        if direct:
            continue
        res = (wet, hyd)
        for mm, f in enumerate(fields):
            if output_created_here:
                outputArrs[mm][hh, ...] = res[f].reshape(ypts.size, xpts.size)
            else:
                outputArrs[mm][hh, ...] += res[f].reshape(ypts.size, xpts.size)
    if output_created_here:
        return outputArrs


def writeResultsToXarray(datetime, xpts, ypts, zpts, crs, wetDelay, hydroDelay, weather_model_file, out_type):
    """delay.py:329-401.  Returns an xarray.Dataset when xarray is installed, else a DelayCube with the
    same variables (`wet`, `hydro` on (z,y,x); coords x, y, z; attrs)."""
    source = os.path.basename(weather_model_file) if isinstance(weather_model_file, (str, os.PathLike)) else 'in-memory'
    attrs = dict(Conventions='CF-1.7', title='RAiDER geo cube', source=source,
                 history=str(dt.datetime.now(tz=dt.timezone.utc)) + ' RAiDER',
                 description=f'RAiDER geo cube - {out_type}',
                 reference_time=datetime.strftime('%Y%m%dT%H:%M:%S') if hasattr(datetime, 'strftime') else str(datetime))
    degrees = _is_4326(crs)
    try:
        import xarray as xr
    except ImportError:
        attrs['_degrees'] = degrees
        if degrees:
            attrs['_crs_cf'] = dict(grid_mapping_name='latitude_longitude', semi_major_axis=6378137.0, inverse_flattening=298.257223563,
                                    longitude_of_prime_meridian=0.0, geographic_crs_name='WGS 84')
        else:
            tm = _tm_params(crs)
            if tm is not None:                                     # CF-1.7 grid mapping of a transverse-Mercator / UTM output grid
                attrs['_crs_cf'] = dict(grid_mapping_name='transverse_mercator', semi_major_axis=tm['a'],
                                        inverse_flattening=1.0 / (1.0 - np.sqrt(1.0 - tm['es'])) if tm['es'] > 0 else 0.0,
                                        longitude_of_prime_meridian=0.0, latitude_of_projection_origin=tm['lat_0'],
                                        longitude_of_central_meridian=tm['lon_0'], scale_factor_at_central_meridian=tm['k_0'],
                                        false_easting=tm['x_0'], false_northing=tm['y_0'])
            else:
                kind = _builtin_crs(crs)
                if kind is not None and kind[0] == 'cone':        # output grid in a conic model CRS: what CRS.to_cf() writes
                    from .crs import cf_from_crs
                    attrs['_crs_cf'] = cf_from_crs(kind[1])[1]
        return DelayCube(dict(wet=np.asarray(wetDelay), hydro=np.asarray(hydroDelay), x=np.asarray(xpts), y=np.asarray(ypts),
                              z=np.asarray(zpts), crs=np.array(-2147483647)), attrs)
    ds = xr.Dataset(
        data_vars=dict(
            wet=(['z', 'y', 'x'], wetDelay, {'units': 'm', 'description': f'wet {out_type} delay', 'grid_mapping': 'crs'}),
            hydro=(['z', 'y', 'x'], hydroDelay, {'units': 'm', 'description': f'hydrostatic {out_type} delay', 'grid_mapping': 'crs'})),
        coords=dict(x=(['x'], xpts), y=(['y'], ypts), z=(['z'], zpts)), attrs=attrs)
    ds['crs'] = -2147483647
    if pyproj is not None:
        for k, v in pyproj.CRS(crs).to_cf().items():
            ds.crs.attrs[k] = v
    ds.z.attrs.update(axis='Z', units='m', description='height above ellipsoid')
    if degrees:
        ds.y.attrs.update(units='degrees_north', standard_name='latitude', long_name='latitude')
        ds.x.attrs.update(units='degrees_east', standard_name='longitude', long_name='longitude')
    else:
        ds.y.attrs.update(axis='Y', standard_name='projection_y_coordinate', long_name='y-coordinate in projected coordinate system', units='m')
        ds.x.attrs.update(axis='X', standard_name='projection_x_coordinate', long_name='x-coordinate in projected coordinate system', units='m')
    return ds


def _builtin_crs(crs):
    """('geodetic' | 'ecef' | 'tm' | 'cone', parameters) for the CRSs handled on the GPU, or None."""
    e = _epsg(crs)
    if e == 4326:
        return 'geodetic', None
    if e == 4978:
        return 'ecef', None
    tm = _tm_params(crs)
    if tm is not None:
        return 'tm', tm
    lcc = _lcc_params(crs)
    if lcc is not None:
        return 'cone', dict(lcc, proj='lcc')
    st = _stere_params(crs)
    if st is not None:
        return 'cone', dict(st, proj='stere')
    return None


def transformPoints(lats, lons, hgts, old_proj, new_proj):
    """delay.py:404-436: (lat, lon, h) in `old_proj` -> stacked (y, x, z) in `new_proj` (`lats` carries y, `lons` carries x).
    EPSG:4326, EPSG:4978, transverse-Mercator CRSs (every UTM zone; national TM grids as a PROJ string / dict) and the conic model
    CRSs (Lambert conformal conic: HRRR; polar stereographic: HRRR-AK) are converted on the GPU in any combination, through
    geodetic coordinates; anything else needs pyproj (as in the reference)."""
    eo, en = _epsg(old_proj), _epsg(new_proj)
    lats, lons, hgts = np.broadcast_arrays(np.asarray(lats, dtype=np.float64), np.asarray(lons, dtype=np.float64),
                                           np.asarray(hgts, dtype=np.float64))
    if (eo is not None and eo == en) or (eo is None and en is None and old_proj == new_proj):
        return np.stack([lats, lons, hgts], axis=-1)
    src, dst = _builtin_crs(old_proj), _builtin_crs(new_proj)
    if src is not None and dst is not None:
        from .utilFcns import conic, transverse_mercator
        # to geodetic (lat, lon, h)
        if src[0] == 'ecef':
            lon, lat, h = ecef2lla(lons, lats, hgts)      # always_xy: x = "lons" argument, y = "lats" argument
        elif src[0] == 'tm':
            lat, lon = transverse_mercator(lats, lons, src[1], inverse=True); h = hgts
        elif src[0] == 'cone':
            lat, lon = conic(lats, lons, src[1], inverse=True); h = hgts
        else:
            lat, lon, h = lats, lons, hgts
        # from geodetic
        if dst[0] == 'ecef':
            x, y, z = lla2ecef(lat, lon, h)
            return np.stack([y, x, z], axis=-1)
        if dst[0] == 'tm':
            y, x = transverse_mercator(lat, lon, dst[1])
            return np.stack([y, x, h], axis=-1)
        if dst[0] == 'cone':
            y, x = conic(lat, lon, dst[1])
            return np.stack([y, x, h], axis=-1)
        return np.stack([lat, lon, h], axis=-1)
    if pyproj is None:
        raise NotImplementedError(f'transformPoints {old_proj} -> {new_proj} needs pyproj (built in: EPSG:4326, EPSG:4978, transverse Mercator / '
                                  'UTM, Lambert conformal conic and polar stereographic, in any combination)')
    t = pyproj.Transformer.from_crs(old_proj, new_proj, always_xy=True)
    res = t.transform(lons, lats, hgts)
    return np.stack([res[1], res[0], res[2]], axis=-1)

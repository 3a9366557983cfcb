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

    def to_netcdf(self, path, time=None, model_name='ERA-5', format='NETCDF4'):
        """The processed weather-model file of WeatherModel.write (weatherModel.py:659-724): dims z, y, x; variables wet, hydro
        (f32), wet_total, hydro_total (f64) - and t, p, e (f32) when the model was produced with return_state=True - with the
        reference's units / standard_name / grid_mapping attributes, 2-D latitude / longitude, the `proj` grid-mapping variable
        carrying `crs_wkt`, and the global attributes.  format='NETCDF4' (what the reference writes through xarray): HDF5 via
        raider_amd.h5write, coordinate variables typed as in the reference's files (x, y f32; z f64); 'NETCDF3_64BIT': classic
        format through scipy.  Either is read back by tropo_delay / getInterpolators, xarray and netCDF4."""
        import datetime as dt
        from .delay import _builtin_crs
        kind = _builtin_crs(self.proj)
        if kind is None or kind[0] not in ('geodetic', 'cone'):
            raise NotImplementedError(f'ProcessedModel.to_netcdf: the model CRS {self.proj!r} has no CF description here (EPSG:4326, Lambert '
                                      'conformal conic and polar stereographic models do)')
        ys, xs, zs = self.pointwise.grid
        wet, hyd = self.pointwise.read(); wt, ht = self.total.read()                     # (y, x, z)
        zyx = lambda v: np.ascontiguousarray(np.asarray(v).transpose(2, 0, 1))
        fields = [('wet', wet, 'f4', 'dimentionless', 'wet_refractivity'), ('hydro', hyd, 'f4', 'dimentionless', 'hydrostatic_refractivity'),
                  ('wet_total', wt, 'f8', 'm', 'total_wet_refractivity'), ('hydro_total', ht, 'f8', 'm', 'total_hydrostatic_refractivity')]
        if self.t is not None:
            fields = [('t', self.t, 'f4', 'K', 'temperature'), ('p', self.p, 'f4', 'Pa', 'pressure'), ('e', self.e, 'f4', 'Pa', 'humidity')] + fields
        gattrs = dict(Conventions='CF-1.6', title='Weather model data and delay calculations', model_name=str(model_name))
        if time is not None:
            gattrs['datetime'] = time.strftime('%Y_%m_%dT%H_%M_%S')
        gattrs['date_created'] = dt.datetime.now().strftime('%Y_%m_%dT%H_%M_%S')
        # EPSG:4326 as current PROJ releases spell it in WKT2:2019 (what pyproj's CRS.to_cf() puts into the reference's files)
        crs_wkt = ('GEOGCRS["WGS 84",ENSEMBLE["World Geodetic System 1984 ensemble",MEMBER["World Geodetic System 1984 (Transit)"],'
                   'MEMBER["World Geodetic System 1984 (G730)"],MEMBER["World Geodetic System 1984 (G873)"],MEMBER["World Geodetic System 1984 (G1150)"],'
                   'MEMBER["World Geodetic System 1984 (G1674)"],MEMBER["World Geodetic System 1984 (G1762)"],MEMBER["World Geodetic System 1984 (G2139)"],'
                   'ELLIPSOID["WGS 84",6378137,298.257223563,LENGTHUNIT["metre",1]],ENSEMBLEACCURACY[2.0]],PRIMEM["Greenwich",0,'
                   'ANGLEUNIT["degree",0.0174532925199433]],CS[ellipsoidal,2],AXIS["geodetic latitude (Lat)",north,ORDER[1],'
                   'ANGLEUNIT["degree",0.0174532925199433]],AXIS["geodetic longitude (Lon)",east,ORDER[2],ANGLEUNIT["degree",0.0174532925199433]],'
                   'USAGE[SCOPE["Horizontal component of 3D system."],AREA["World."],BBOX[-90,-180,90,180]],ID["EPSG",4326]]')
        proj_attrs = dict(crs_wkt=crs_wkt, semi_major_axis=6378137.0, semi_minor_axis=6356752.314245179, inverse_flattening=298.257223563,
                          reference_ellipsoid_name='WGS 84', longitude_of_prime_meridian=0.0, prime_meridian_name='Greenwich',
                          geographic_crs_name='WGS 84', horizontal_datum_name='World Geodetic System 1984 ensemble', grid_mapping_name='latitude_longitude',
                          grid_mapping='proj')      # (weatherModel.py:716-717 tags every data variable, `proj` included)
        lon2, lat2 = np.meshgrid(xs, ys)
        if kind[0] == 'cone':                                   # projected model (HRRR, HRRR-AK): what CRS.to_cf() writes; geodetic 2-D coordinates
            from .crs import cf_from_crs
            from .utilFcns import conic
            _, cf = cf_from_crs(kind[1])
            proj_attrs = dict(cf, grid_mapping='proj')
            lat2, lon2 = conic(lat2, lon2, kind[1], inverse=True)          # (y, x) metres -> (lat, lon) degrees, on the device
        if format.upper().startswith('NETCDF4'):
            from .h5write import write_netcdf4
            # (the reference's files carry x / y as float32 because its ERA-5 axes ARE float32; axes that are not exactly
            # representable in float32 stay float64 - a rounded axis would move every interpolation weight)
            f32_exact = lambda a: bool(np.array_equal(np.asarray(a, np.float32).astype(np.float64), np.asarray(a, np.float64)))
            hdt = np.float32 if (f32_exact(xs) and f32_exact(ys)) else np.float64
            variables = {'z': (('z',), np.asarray(zs, np.float64), {}), 'y': (('y',), np.asarray(ys, hdt), {}), 'x': (('x',), np.asarray(xs, hdt), {}),
                         'latitude': (('y', 'x'), lat2.astype(hdt), {}), 'longitude': (('y', 'x'), lon2.astype(hdt), {})}
            for name, arr, typ, units, std in fields:
                variables[name] = (('z', 'y', 'x'), zyx(arr).astype(typ), dict(units=units, standard_name=std, grid_mapping='proj', coordinates='latitude longitude'))
            variables['proj'] = ((), np.array(0, dtype=np.int64), proj_attrs)
            write_netcdf4(path, dict(z=np.size(zs), y=np.size(ys), x=np.size(xs)), variables, gattrs)
            return str(path)
        from scipy.io import netcdf_file
        with netcdf_file(str(path), 'w', version=2) as f:
            for k, v in gattrs.items():
                setattr(f, k, v)
            for d, v in (('z', zs), ('y', ys), ('x', xs)):
                f.createDimension(d, int(np.size(v)))
                f.createVariable(d, 'f8', (d,))[:] = np.asarray(v, dtype=np.float64)
            f.createVariable('latitude', 'f8', ('y', 'x'))[:] = lat2
            f.createVariable('longitude', 'f8', ('y', 'x'))[:] = lon2
            for name, arr, typ, units, std in fields:
                v = f.createVariable(name, typ, ('z', 'y', 'x'))
                v[:] = zyx(arr)
                v.units = units; v.standard_name = std; v.grid_mapping = 'proj'
            pj = f.createVariable('proj', 'i4', ())
            pj.data[()] = 0
            for k, v in proj_attrs.items():
                setattr(pj, k, np.float64(v) if isinstance(v, float) else v)      # (a bare Python float is narrowed to float32 by scipy's writer)
        return str(path)

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


# ------------------------------------------------------------------------------------------------
# ECMWF hybrid model levels (ERA-5 / HRES raw files) -> the producer's inputs
# ------------------------------------------------------------------------------------------------
def ecmwf_l137():
    """ECMWF's L137 hybrid coefficients and the 145 heights ECMWF models are resampled to (models/ecmwf.py:42-46 with the
    tables of models/model_levels.py, shipped as data): dict(a[138], b[138], level_heights[145] descending)."""
    from pathlib import Path
    d = np.load(Path(__file__).resolve().parent / 'data' / 'ecmwf_l137.npz')
    return {k: d[k] for k in d.files}


def read_ecmwf_model_level_file(path, ll_bounds=None):
    """Raw ERA-5 / HRES model-level file as the CDS / MARS write it (NetCDF-3, packed int16 `z, t, q, lnsp` on
    (time, level, latitude, longitude)) -> dict(lats, lons, z, lnsp (ny, nx), t, q (nlev, ny, nx)), float32, latitude and
    longitude ascending, longitudes in [-180, 180), cut to ll_bounds = (S, N, W, E).  Mirrors ECMWF._makeDataCubes and the
    flips of _load_model_level (models/ecmwf.py:305-337, 58-79); host-side I/O (scipy), no GPU work."""
    from scipy.io import netcdf_file
    with netcdf_file(str(path), 'r', mmap=False) as f:
        def decode(name):                     # CF scale/offset decoding in float32, as xarray does for int16 data
            v = f.variables[name]
            raw = np.array(v.data)
            out = raw.astype(np.float32)
            out *= np.float32(getattr(v, 'scale_factor', 1.0))
            out += np.float32(getattr(v, 'add_offset', 0.0))
            fill = getattr(v, '_FillValue', None)
            if fill is not None:
                out[raw == fill] = np.nan
            return np.squeeze(out)
        z, t, q, lnsp = decode('z'), decode('t'), decode('q'), decode('lnsp')
        lats = np.array(f.variables['latitude'].data, dtype=np.float32)
        lons = np.array(f.variables['longitude'].data, dtype=np.float32)
    lons = ((lons + 180) % 360) - 180
    z, lnsp = z[0], lnsp[0]                   # the two surface fields are stored on level 1
    if ll_bounds is not None:
        S, N, W, E = ll_bounds
        my, mx = (S <= lats) & (N >= lats), (W <= lons) & (E >= lons)
        lats, lons = lats[my], lons[mx]
        z, lnsp, t, q = z[my][:, mx], lnsp[my][:, mx], t[:, my][:, :, mx], q[:, my][:, :, mx]
    if z.size == 0:
        raise RuntimeError('There is no data in z, you may have a problem with your mask')           # ecmwf.py:334-335
    if lats.size > 1 and lats[0] > lats[1]:
        z, lnsp, t, q, lats = z[::-1], lnsp[::-1], t[:, ::-1], q[:, ::-1], lats[::-1]
    if lons.size > 1 and lons[0] > lons[1]:
        z, lnsp, t, q, lons = z[..., ::-1], lnsp[..., ::-1], t[..., ::-1], q[..., ::-1], lons[::-1]
    c = np.ascontiguousarray
    return dict(lats=c(lats), lons=c(lons), z=c(z), lnsp=c(lnsp), t=c(t), q=c(q))


def ecmwf_model_levels(z_surf, lnsp, t, q, lats, a=None, b=None, R_d=287.06, ctx=None):
    """utilFcns.calcgeoh + geo_to_ht + the re-ordering of ecmwf.py:92-110 on the GPU (rdr_ecmwf_model_levels): surface
    geopotential and log surface pressure (ny, nx), temperature and specific humidity (nlev, ny, nx) with level 1 = model top,
    latitudes (ny,) -> (p, zs), both (ny, nx, nlev) float64 with the bottom level first (the producer's layout).  NumPy in ->
    NumPy out; torch tensors on the GPU in -> tensors out."""
    ctx = ctx or Context.default()
    if a is None or b is None:
        tab = ecmwf_l137()
        a, b = tab['a'], tab['b']
    a, b = f64(a), f64(b)
    dev = _is_dev(t)
    lats = np.ascontiguousarray(lats.cpu().numpy() if _is_dev(lats) else lats, dtype=np.float32)
    if dev:
        import torch
        zt, lt, tt, qt = (v.to(torch.float32).contiguous() for v in (z_surf, lnsp, t, q))
        ctx.adopt_torch_stream(tt)
        nlev, ny, nx = (int(v) for v in tt.shape)
        p = torch.empty((ny, nx, nlev), dtype=torch.float64, device=tt.device); zs = torch.empty_like(p)
    else:
        zt, lt, tt, qt = (np.ascontiguousarray(v, dtype=np.float32) for v in (z_surf, lnsp, t, q))
        nlev, ny, nx = tt.shape
        p = np.empty((ny, nx, nlev)); zs = np.empty_like(p)
    if tuple(qt.shape) != (nlev, ny, nx) or tuple(zt.shape) != (ny, nx) or tuple(lt.shape) != (ny, nx) or lats.size != ny:
        raise ValueError('ecmwf_model_levels: t, q must be (nlev, ny, nx); z_surf, lnsp (ny, nx); lats (ny,)')
    if a.size != nlev + 1 or b.size != nlev + 1:
        raise ValueError(f'I have here a model with {nlev} levels, but parameters a and b have lengths {a.size} and {b.size} '
                         'respectively. Of course, these three numbers should be equal.')                   # utilFcns.py:813-817
    check(ctx.lib.rdr_ecmwf_model_levels(ctx.handle, ptr(zt), ptr(lt), ptr(tt), ptr(qt), ptr(lats), ptr(a), ptr(b), nlev, ny, nx, float(R_d),
                                         ptr(p), ptr(zs), L.RDR_DEVICE if dev else L.RDR_HOST), ctx.handle)
    return p, zs


def load_ecmwf_model_levels(path, ll_bounds=None, new_z=None, return_state=False, ctx=None):
    """WeatherModel.load for a raw ERA-5 / HRES model-level file (weatherModel.py:235-262 with ECMWF._load_model_level): file ->
    hybrid-level pressures and geometric heights -> e -> uniform z levels -> NaN fill -> refractivities -> padded bottom level ->
    ZTDs, everything after the file read on the GPU.  Returns the ProcessedModel that tropo_delay / getInterpolators take."""
    ctx = ctx or Context.default()
    raw = read_ecmwf_model_level_file(path, ll_bounds)
    tab = ecmwf_l137()
    p, zs = ecmwf_model_levels(raw['z'], raw['lnsp'], raw['t'], raw['q'], raw['lats'], tab['a'], tab['b'], ctx=ctx)
    up = lambda v: np.ascontiguousarray(np.flip(v.transpose(1, 2, 0), axis=2), dtype=np.float64)          # ecmwf.py:98-106
    if new_z is None:
        new_z = np.flipud(tab['level_heights'])                                                            # ecmwf.py:44
    return cubes_from_model_levels(raw['lons'].astype(np.float64), raw['lats'].astype(np.float64), zs, p, up(raw['t']), up(raw['q']), 'q',
                                   new_z=new_z, return_state=return_state, ctx=ctx)

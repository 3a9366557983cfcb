"""Geodesy helpers of the delay path (tools/RAiDER/utilFcns.py:55-137).

lla2ecef / ecef2lla run on the GPU (the reference goes through pyproj/PROJ, utilFcns.py:77-88);
the ENU rotations and degree trig are the same few NumPy expressions the reference uses and are kept
on the host because the reference's callers use them on tiny arrays outside the hot loops (inside the
ray kernels the same conversions are fused on-device, csrc/geodesy.h)."""

import numpy as np

from ._lib import Context, check, f64, ptr, RDR_HOST


def projectDelays(delay, inc):
    """utilFcns.py:55-59."""
    if np.any(np.asarray(inc) == 90):
        raise ZeroDivisionError
    return delay / cosd(inc)


def sind(x):
    """utilFcns.py:67-69."""
    return np.sin(np.radians(x))


def cosd(x):
    """utilFcns.py:72-74."""
    return np.cos(np.radians(x))


def lla2ecef(lat, lon, height):
    """utilFcns.py:77-81 - returns the (x, y, z) tuple pyproj's always_xy transform returns."""
    lat, lon, height = np.broadcast_arrays(np.asarray(lat, dtype=np.float64), np.asarray(lon, dtype=np.float64),
                                           np.asarray(height, dtype=np.float64))
    shp = lat.shape
    ctx = Context.default()
    la, lo, hh = f64(lat).ravel(), f64(lon).ravel(), f64(height).ravel()
    xyz = np.empty((la.size, 3))
    check(ctx.lib.rdr_lla2ecef(ctx.handle, ptr(la), ptr(lo), ptr(hh), la.size, ptr(xyz), RDR_HOST), ctx.handle)
    return xyz[:, 0].reshape(shp), xyz[:, 1].reshape(shp), xyz[:, 2].reshape(shp)


def ecef2lla(x, y, z):
    """utilFcns.py:84-88 - returns (lon, lat, height) (always_xy order)."""
    x, y, z = np.broadcast_arrays(np.asarray(x, dtype=np.float64), np.asarray(y, dtype=np.float64), np.asarray(z, dtype=np.float64))
    shp = x.shape
    xyz = np.ascontiguousarray(np.stack([x.ravel(), y.ravel(), z.ravel()], axis=-1))
    ctx = Context.default()
    n = xyz.shape[0]
    lon, lat, h = np.empty(n), np.empty(n), np.empty(n)
    check(ctx.lib.rdr_ecef2lla(ctx.handle, ptr(xyz), n, ptr(lon), ptr(lat), ptr(h), RDR_HOST), ctx.handle)
    return lon.reshape(shp), lat.reshape(shp), h.reshape(shp)


def enu2ecef(east, north, up, lat0, lon0, h0):
    """utilFcns.py:91-121: rotate a local ENU vector at (lat0, lon0) into ECEF; h0 is unused there too."""
    slat, clat, slon, clon = sind(lat0), cosd(lat0), sind(lon0), cosd(lon0)
    t = clat * up - slat * north
    w = slat * up + clat * north
    u = clon * t - slon * east
    v = slon * t + clon * east
    return np.stack((u, v, w), axis=-1)


def ecef2enu(xyz, lat, lon, height):
    """utilFcns.py:124-137."""
    x, y, z = xyz[..., 0], xyz[..., 1], xyz[..., 2]
    slat, clat, slon, clon = sind(lat), cosd(lat), sind(lon), cosd(lon)
    t = clon * x + slon * y
    e = -slon * x + clon * y
    n = -slat * t + clat * z
    u = clat * t + slat * z
    return np.stack((e, n, u), axis=-1)


def utm_params(epsg):
    """Transverse-Mercator parameters of a WGS 84 / UTM zone EPSG code (326zz north, 327zz south), or None."""
    epsg = int(epsg)
    if 32601 <= epsg <= 32660 or 32701 <= epsg <= 32760:
        zone = epsg % 100
        return dict(a=6378137.0, es=0.0066943799901413165, lat_0=0.0, lon_0=6.0 * zone - 183.0, k_0=0.9996, x_0=500000.0,
                    y_0=10000000.0 if epsg >= 32701 else 0.0)
    return None


def conic(a_in, b_in, params, inverse=False):
    """Forward: (lat, lon) deg -> (y, x) m; inverse: (y, x) m -> (lat, lon) deg for a conic model CRS, on the GPU (rdr_transform_cone).
    params: dict(proj='lcc', a, es, lat_1, lat_2, lat_0, lon_0, x_0, y_0) (HRRR, models/hrrr.py:248-259) or
    dict(proj='stere', a, es, lat_0, lat_ts, k_0, lon_0, x_0, y_0) (HRRR-AK, models/hrrr.py:22-25; polar aspect only)."""
    from . import _lib as L
    from ._lib import Context, check, f64, ptr
    a_in, b_in = np.broadcast_arrays(np.asarray(a_in, dtype=np.float64), np.asarray(b_in, dtype=np.float64))
    shp = a_in.shape
    ua, ub = f64(a_in).ravel(), f64(b_in).ravel()
    oa, ob = np.empty(ua.size), np.empty(ua.size)
    if params.get('proj', 'lcc') == 'stere':
        kind = 2
        lat_ts = params.get('lat_ts')
        p = np.array([params['a'], params['es'], params['lat_0'], np.nan if lat_ts is None else lat_ts, params.get('k_0', 1.0), params['lon_0'],
                      params.get('x_0', 0.0), params.get('y_0', 0.0)], dtype=np.float64)
    else:
        kind = 1
        p = np.array([params['a'], params['es'], params['lat_1'], params['lat_2'], params['lat_0'], params['lon_0'], params.get('x_0', 0.0),
                      params.get('y_0', 0.0)], dtype=np.float64)
    ctx = Context.default()
    check(ctx.lib.rdr_transform_cone(ctx.handle, kind, ptr(p), p.size, int(bool(inverse)), ptr(ua), ptr(ub), ua.size, ptr(oa), ptr(ob), L.RDR_HOST),
          ctx.handle)
    return oa.reshape(shp), ob.reshape(shp)


def transverse_mercator(a_in, b_in, params, inverse=False):
    """Forward: (lat, lon) deg -> (y, x) m; inverse: (y, x) m -> (lat, lon) deg, on the GPU (rdr_transform_tm: Krueger series to n^6,
    the formulation of PROJ's etmerc / utm).  params: dict(a, es, lat_0, lon_0, k_0, x_0, y_0)."""
    import ctypes as C
    from . import _lib as L
    from ._lib import Context, check, f64, ptr
    a_in, b_in = np.broadcast_arrays(np.asarray(a_in, dtype=np.float64), np.asarray(b_in, dtype=np.float64))
    shp = a_in.shape
    ua, ub = f64(a_in).ravel(), f64(b_in).ravel()
    oa, ob = np.empty(ua.size), np.empty(ua.size)
    p = np.array([params[k] for k in ('a', 'es', 'lat_0', 'lon_0', 'k_0', 'x_0', 'y_0')], dtype=np.float64)
    ctx = Context.default()
    check(ctx.lib.rdr_transform_tm(ctx.handle, ptr(p), p.size, int(bool(inverse)), ptr(ua), ptr(ub), ua.size, ptr(oa), ptr(ob), L.RDR_HOST), ctx.handle)
    return oa.reshape(shp), ob.reshape(shp)


def rio_open(path, userNDV=None, band=None):
    """utilFcns.py:164-202 (rasterio when installed, else flat-binary rasters with a .vrt / ENVI .hdr side-car)."""
    from .rawraster import rio_open as _rio_open
    return _rio_open(path, userNDV=userNDV, band=band)


def writeArrayToRaster(array, path, noDataValue=0.0, fmt='ENVI', proj=None, gt=None):
    """utilFcns.py:257-304: a 2-D array as a GDAL-readable raster (float -> float32, complex -> complex64, anything else -> uint8).
    rasterio writes it when installed; without it the ENVI format is written natively (flat binary + .hdr), other formats raise."""
    from pathlib import Path
    array = np.asarray(array)
    if array.ndim != 2:
        raise RuntimeError(f'writeArrayToRaster: cannot write an array of shape {np.shape(array)} to a raster image')
    dtype = np.complex64 if 'complex' in str(array.dtype) else (np.float32 if 'float' in str(array.dtype) else np.uint8)
    path = Path(path)
    if fmt == 'nc':
        fmt = 'GTiff'; path = path.with_suffix('.tif')
    try:
        import rasterio
    except ImportError:
        rasterio = None
    if rasterio is not None:
        trans = None
        if gt is not None:
            try:
                trans = rasterio.Affine.from_gdal(*gt)
            except TypeError:
                trans = gt
        with rasterio.open(path, mode='w', count=1, width=array.shape[1], height=array.shape[0], dtype=dtype, crs=proj, nodata=noDataValue, driver=fmt,
                           transform=trans) as dst:
            dst.write(array.astype(dtype), 1)
        return
    if str(fmt).upper() != 'ENVI':
        raise ImportError(f'writing {fmt} rasters needs rasterio, which is not installed (ENVI is written without it)')
    from .rawraster import write_envi
    write_envi(array.astype(dtype), path, nodata=noDataValue, geotransform=gt, proj=proj)


def writeDelays(aoi, wetDelay, hydroDelay, wet_path, hydro_path=None, outformat=None, ndv=0.0):
    """utilFcns.py:431-464: station AOIs -> the station CSV with wetDelay / hydroDelay / totalDelay columns; raster AOIs -> two rasters."""
    from pathlib import Path
    import pandas as pd
    wetDelay[np.isnan(wetDelay)] = ndv                  # (in place, as the reference does)
    hydroDelay[np.isnan(hydroDelay)] = ndv
    t = aoi.type() if callable(getattr(aoi, 'type', None)) else getattr(aoi, 'type', None)
    if t == 'station_file':
        df = pd.read_csv(aoi._filename).drop_duplicates(subset=['Lat', 'Lon'])
        df['wetDelay'] = wetDelay
        df['hydroDelay'] = hydroDelay
        df['totalDelay'] = wetDelay + hydroDelay
        df.to_csv(str(wet_path), index=False)
        return
    if hydro_path is None:
        raise ValueError('Hydro delay file path must be specified if the AOI is not a station file')
    proj = aoi.projection() if hasattr(aoi, 'projection') else None
    gt = aoi.geotransform() if hasattr(aoi, 'geotransform') else None
    writeArrayToRaster(wetDelay, Path(wet_path), noDataValue=ndv, fmt=outformat or 'ENVI', proj=proj, gt=gt)
    writeArrayToRaster(hydroDelay, Path(hydro_path), noDataValue=ndv, fmt=outformat or 'ENVI', proj=proj, gt=gt)

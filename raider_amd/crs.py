"""The CRS of a weather-model / delay-cube file without pyproj.

tropo_delay reads the model CRS off the file's `proj` variable: `CRS.from_wkt(ds['proj'].attrs['crs_wkt'])` (delay.py:66-73); the
reference writes that variable with `pyproj.CRS.to_cf()` (models/weatherModel.py:711-715), i.e. the WKT **and** the CF-1.x grid-mapping
attributes (`grid_mapping_name`, `standard_parallel`, `longitude_of_central_meridian`, `semi_major_axis`, ...).  pyproj is used
when it is installed.  Otherwise the CRS is taken from the CF attributes, or - for files that only carry `crs_wkt` - from a small
WKT reader (WKT2:2019 as current PROJ writes it, and the older WKT1 / ESRI spellings).  What comes out is what the rest of the
package takes as a CRS: the int 4326 / an EPSG code, or a PROJ-style dict (`proj`, `lat_1`, `lon_0`, `a`, `rf` ...) for the
projections built into the kernels (Lambert conformal conic: HRRR; polar stereographic: HRRR-AK; transverse Mercator)."""
import re

import numpy as np

_WGS84_A, _WGS84_RF = 6378137.0, 298.257223563


def _is_wgs84(a, rf):
    """WGS 84 (or GRS 80, 0.1 mm apart) within what a float32 attribute keeps (a NetCDF-3 writer may store 298.257223563 as 298.25723)."""
    return abs(a - _WGS84_A) < 0.5 and abs(rf - _WGS84_RF) < 1e-4


# ---- CF grid-mapping attributes ---------------------------------------------------------------------------------------------
def _scalar(v):
    return float(np.asarray(v).ravel()[0])


def _ellipsoid_from_cf(at):
    if 'semi_major_axis' in at:
        a = _scalar(at['semi_major_axis'])
    elif 'earth_radius' in at:
        return dict(a=_scalar(at['earth_radius']), rf=0.0)
    else:
        return dict(a=_WGS84_A, rf=_WGS84_RF)
    if 'inverse_flattening' in at:
        return dict(a=a, rf=_scalar(at['inverse_flattening']))
    if 'semi_minor_axis' in at:
        b = _scalar(at['semi_minor_axis'])
        return dict(a=a, rf=0.0 if b == a else a / (a - b))
    return dict(a=a, rf=0.0)


def _with_ellipsoid(d, ell):
    d['a'] = ell['a']
    if ell['rf']:
        d['rf'] = ell['rf']
    else:
        d['b'] = ell['a']
    return d


def crs_from_cf(attrs):
    """CF grid-mapping attributes -> 4326 or a PROJ-style dict; None when `grid_mapping_name` is absent or not one of the built-in
    projections."""
    name = attrs.get('grid_mapping_name')
    if name is None:
        return None
    name = str(name)
    ell = _ellipsoid_from_cf(attrs)
    fe, fn = _scalar(attrs.get('false_easting', 0.0)), _scalar(attrs.get('false_northing', 0.0))
    if name == 'latitude_longitude':
        if _is_wgs84(ell['a'], ell['rf']):
            return 4326
        return _with_ellipsoid(dict(proj='longlat'), ell)
    if name == 'lambert_conformal_conic':
        sp = np.atleast_1d(np.asarray(attrs['standard_parallel'], dtype=np.float64))
        lat_0 = _scalar(attrs.get('latitude_of_projection_origin', sp[0]))
        return _with_ellipsoid(dict(proj='lcc', lat_1=float(sp[0]), lat_2=float(sp[-1]), lat_0=lat_0,
                                    lon_0=_scalar(attrs.get('longitude_of_central_meridian', 0.0)), x_0=fe, y_0=fn), ell)
    if name == 'polar_stereographic':
        d = dict(proj='stere', lat_0=_scalar(attrs.get('latitude_of_projection_origin', 90.0)),
                 lon_0=_scalar(attrs.get('straight_vertical_longitude_from_pole', attrs.get('longitude_of_projection_origin', 0.0))), x_0=fe, y_0=fn)
        if 'standard_parallel' in attrs:
            d['lat_ts'] = _scalar(attrs['standard_parallel'])
        else:
            d['k_0'] = _scalar(attrs.get('scale_factor_at_projection_origin', 1.0))
        return _with_ellipsoid(d, ell)
    if name == 'transverse_mercator':
        return _with_ellipsoid(dict(proj='tmerc', lat_0=_scalar(attrs.get('latitude_of_projection_origin', 0.0)),
                                    lon_0=_scalar(attrs.get('longitude_of_central_meridian', 0.0)),
                                    k_0=_scalar(attrs.get('scale_factor_at_central_meridian', 1.0)), x_0=fe, y_0=fn), ell)
    return None


# ---- WKT ------------------------------------------------------------------------------------------------------------------
_TOKEN = re.compile(r'\s*(?:"((?:[^"]|"")*)"|([A-Za-z_][A-Za-z_0-9]*)|([-+]?(?:\d+\.?\d*|\.\d+)(?:[eE][-+]?\d+)?)|([\[\](),]))')


def parse_wkt(wkt):
    """WKT text -> nested nodes (keyword, [arguments]); an argument is a str, a float or another node."""
    pos = 0
    n = len(wkt)

    def token():
        nonlocal pos
        m = _TOKEN.match(wkt, pos)
        if not m:
            if wkt[pos:].strip() == '':
                return None
            raise ValueError(f'cannot parse WKT at position {pos}: {wkt[pos:pos + 30]!r}')
        pos = m.end()
        if m.group(1) is not None:
            return ('str', m.group(1).replace('""', '"'))
        if m.group(2) is not None:
            return ('kw', m.group(2))
        if m.group(3) is not None:
            return ('num', float(m.group(3)))
        return ('p', m.group(4))

    def node(kw):
        nonlocal pos
        args = []
        while True:
            t = token()
            if t is None:
                raise ValueError('unterminated WKT')
            if t[0] == 'p' and t[1] in '])':
                return (kw.upper(), args)
            if t[0] == 'p' and t[1] == ',':
                continue
            if t[0] == 'kw':
                save = pos
                t2 = token()
                if t2 is not None and t2[0] == 'p' and t2[1] in '[(':
                    args.append(node(t[1]))
                else:
                    pos = save
                    args.append(t[1])            # a bare enumeration value (axis direction, CS type)
            else:
                args.append(t[1])

    t = token()
    if t is None or t[0] != 'kw':
        raise ValueError('WKT does not start with a keyword')
    t2 = token()
    if t2 is None or t2[0] != 'p' or t2[1] not in '[(':
        raise ValueError('WKT keyword without arguments')
    out = node(t[1])
    if pos < n and wkt[pos:].strip():
        raise ValueError('trailing text after the WKT')
    return out


def _children(nd, *kws):
    return [a for a in nd[1] if isinstance(a, tuple) and a[0] in kws]


def _find(nd, *kws):
    for a in nd[1]:
        if isinstance(a, tuple):
            if a[0] in kws:
                return a
            r = _find(a, *kws)
            if r is not None:
                return r
    return None


def _epsg_id(nd):
    """The node's OWN identifier (ID["EPSG", n] / AUTHORITY["EPSG","n"]), not a child's."""
    for a in _children(nd, 'ID', 'AUTHORITY'):
        if len(a[1]) >= 2 and str(a[1][0]).upper() == 'EPSG':
            try:
                return int(float(a[1][1]))
            except (TypeError, ValueError):
                return None
    return None


_PARAM_ALIASES = {
    'lat_0': ('latitude of false origin', 'latitude of natural origin', 'latitude_of_origin', 'latitude_of_center', 'latitude of projection centre'),
    'lon_0': ('longitude of false origin', 'longitude of natural origin', 'central_meridian', 'longitude_of_center', 'longitude of origin',
              'longitude_of_origin', 'straight_vertical_longitude_from_pole'),
    'lat_1': ('latitude of 1st standard parallel', 'standard_parallel_1'),
    'lat_2': ('latitude of 2nd standard parallel', 'standard_parallel_2'),
    'lat_ts': ('latitude of standard parallel', 'standard_parallel_1'),
    'k_0': ('scale factor at natural origin', 'scale_factor', 'scale factor at projection centre'),
    'x_0': ('easting at false origin', 'false easting', 'false_easting', 'easting at projection centre'),
    'y_0': ('northing at false origin', 'false northing', 'false_northing', 'northing at projection centre'),
}


def crs_from_wkt(wkt):
    """WKT (2 or 1) -> an EPSG int when the CRS itself carries an EPSG identifier, 4326 for a WGS 84 geographic CRS, else a PROJ-style
    dict for the built-in projections.  Raises ValueError for anything else."""
    root = parse_wkt(wkt)
    kind = root[0]
    code = _epsg_id(root)
    if code is not None:
        return code
    ell = _find(root, 'ELLIPSOID', 'SPHEROID')
    if ell is None or len(ell[1]) < 3:
        raise ValueError('WKT without an ellipsoid')
    a, rf = float(ell[1][1]), float(ell[1][2])
    lu = _children(ell, 'LENGTHUNIT', 'UNIT')
    if lu and len(lu[0][1]) >= 2:
        a *= float(lu[0][1][1])
    ellips = dict(a=a, rf=rf)
    if kind in ('GEOGCRS', 'GEOGRAPHICCRS', 'GEOGCS', 'GEODCRS', 'GEODETICCRS', 'BASEGEOGCRS'):
        if _is_wgs84(a, rf):
            return 4326
        return _with_ellipsoid(dict(proj='longlat'), ellips)
    if kind not in ('PROJCRS', 'PROJECTEDCRS', 'PROJCS'):
        raise ValueError(f'unsupported WKT root {kind}')
    conv = _find(root, 'CONVERSION')
    holder = conv if conv is not None else root
    meth = _find(holder, 'METHOD', 'PROJECTION')
    if meth is None:
        raise ValueError('projected WKT without a METHOD / PROJECTION')
    mname = str(meth[1][0]).lower().replace('_', ' ')
    params = {}
    for p in _children(holder, 'PARAMETER'):
        if len(p[1]) >= 2:
            val = float(p[1][1])
            for u in _children(p, 'ANGLEUNIT', 'LENGTHUNIT'):              # WKT2 parameters carry their unit (WKT1 ones do not)
                if len(u[1]) >= 2:
                    f = float(u[1][1])
                    val *= f if u[0] == 'LENGTHUNIT' else f / 0.0174532925199433      # metres / degrees
            params[str(p[1][0]).lower()] = val
    if kind == 'PROJCS':                                      # WKT1: linear parameters are in the CRS's own unit
        u = _children(root, 'UNIT')
        if u and len(u[0][1]) >= 2 and float(u[0][1][1]) != 1.0:
            for key in list(params):
                if key in _PARAM_ALIASES['x_0'] or key in _PARAM_ALIASES['y_0']:
                    params[key] *= float(u[0][1][1])

    def get(key, default=None):
        for alias in _PARAM_ALIASES[key]:
            if alias in params:
                return params[alias]
        return default
    if 'lambert' in mname and 'conformal' in mname:
        lat_0 = get('lat_0', 0.0)
        if '1sp' in mname:
            d = dict(proj='lcc', lat_1=lat_0, lat_2=lat_0, lat_0=lat_0, lon_0=get('lon_0', 0.0), k_0=get('k_0', 1.0), x_0=get('x_0', 0.0), y_0=get('y_0', 0.0))
        else:
            lat_1 = get('lat_1', lat_0)
            d = dict(proj='lcc', lat_1=lat_1, lat_2=get('lat_2', lat_1), lat_0=lat_0, lon_0=get('lon_0', 0.0), x_0=get('x_0', 0.0), y_0=get('y_0', 0.0))
        return _with_ellipsoid(d, ellips)
    if 'polar stereographic' in mname or mname == 'stereographic north pole' or mname == 'stereographic south pole':
        if 'variant b' in mname:
            ts = get('lat_ts')
            d = dict(proj='stere', lat_0=90.0 if ts >= 0 else -90.0, lat_ts=ts, lon_0=get('lon_0', 0.0), x_0=get('x_0', 0.0), y_0=get('y_0', 0.0))
        elif 'variant a' in mname:
            d = dict(proj='stere', lat_0=get('lat_0', 90.0), k_0=get('k_0', 1.0), lon_0=get('lon_0', 0.0), x_0=get('x_0', 0.0), y_0=get('y_0', 0.0))
        else:                                                 # WKT1 Polar_Stereographic: latitude_of_origin IS the latitude of true scale
            ts = get('lat_0', 90.0)
            if 'south' in mname:
                ts = -abs(ts)
            d = dict(proj='stere', lat_0=90.0 if ts >= 0 else -90.0, lon_0=get('lon_0', 0.0), x_0=get('x_0', 0.0), y_0=get('y_0', 0.0))
            if abs(abs(ts) - 90.0) > 1e-9:
                d['lat_ts'] = ts
            else:
                d['k_0'] = get('k_0', 1.0)
        return _with_ellipsoid(d, ellips)
    if 'transverse mercator' in mname:
        return _with_ellipsoid(dict(proj='tmerc', lat_0=get('lat_0', 0.0), lon_0=get('lon_0', 0.0), k_0=get('k_0', 1.0), x_0=get('x_0', 0.0),
                                    y_0=get('y_0', 0.0)), ellips)
    raise ValueError(f'projection method {meth[1][0]!r} is not built in (Lambert conformal conic, polar stereographic, transverse Mercator)')


def crs_from_proj_var(attrs):
    """The CRS of a file's grid-mapping variable (`proj` of a processed weather model, `crs` of a delay cube) from its attributes:
    pyproj on `crs_wkt` when it is installed (the reference's route), else the CF attributes, else the WKT reader."""
    wkt = attrs.get('crs_wkt', attrs.get('spatial_ref'))
    if isinstance(wkt, bytes):
        wkt = wkt.decode()
    try:
        import pyproj
        if wkt is not None:
            return pyproj.CRS.from_wkt(wkt)
        return pyproj.CRS.from_cf(dict(attrs))
    except ImportError:
        pass
    crs = crs_from_cf(attrs)
    if crs is not None:
        return crs
    if wkt is None:
        raise KeyError('crs_wkt')
    return crs_from_wkt(wkt)


# ---- the other way: what pyproj's CRS.to_cf() writes, for the built-in projections ---------------------------------------------------
def _angle(v):
    return f'{float(v):.15g}'


def cf_from_crs(d):
    """PROJ-style dict of a built-in projection -> (crs_wkt in WKT2:2019, CF grid-mapping attributes) - what `CRS.to_cf()` gives the
    reference's writer (models/weatherModel.py:711-715).  Names are 'unknown', as PROJ writes them for a CRS made from parameters."""
    from .delay import _ellipsoid
    a, es = _ellipsoid(d)
    rf = 0.0 if es == 0 else 1.0 / (1.0 - np.sqrt(1.0 - es))
    b = a * np.sqrt(1.0 - es)
    deg = 'ANGLEUNIT["degree",0.0174532925199433]'
    met = 'LENGTHUNIT["metre",1]'
    base = (f'BASEGEOGCRS["unknown",DATUM["unknown",ELLIPSOID["unknown",{a:.15g},{rf:.15g},LENGTHUNIT["metre",1,ID["EPSG",9001]]]],'
            f'PRIMEM["Greenwich",0,{deg},ID["EPSG",8901]]]')
    cs = 'CS[Cartesian,2],AXIS["(E)",east,ORDER[1],LENGTHUNIT["metre",1,ID["EPSG",9001]]],AXIS["(N)",north,ORDER[2],LENGTHUNIT["metre",1,ID["EPSG",9001]]]'
    cf = dict(semi_major_axis=a, semi_minor_axis=float(b), inverse_flattening=float(rf), reference_ellipsoid_name='unknown', longitude_of_prime_meridian=0.0,
              prime_meridian_name='Greenwich', geographic_crs_name='unknown', horizontal_datum_name='unknown', projected_crs_name='unknown')
    x0, y0 = float(d.get('x_0', 0.0)), float(d.get('y_0', 0.0))
    kind = d.get('proj')
    if kind == 'lcc':
        lat_1 = float(d.get('lat_1', d.get('lat_0', 0.0))); lat_2 = float(d.get('lat_2', lat_1)); lat_0 = float(d.get('lat_0', 0.0)); lon_0 = float(d.get('lon_0', 0.0))
        conv = (f'CONVERSION["unknown",METHOD["Lambert Conic Conformal (2SP)",ID["EPSG",9802]],PARAMETER["Latitude of false origin",{_angle(lat_0)},{deg},ID["EPSG",8821]],'
                f'PARAMETER["Longitude of false origin",{_angle(lon_0)},{deg},ID["EPSG",8822]],PARAMETER["Latitude of 1st standard parallel",{_angle(lat_1)},{deg},ID["EPSG",8823]],'
                f'PARAMETER["Latitude of 2nd standard parallel",{_angle(lat_2)},{deg},ID["EPSG",8824]],PARAMETER["Easting at false origin",{x0:.15g},{met},ID["EPSG",8826]],'
                f'PARAMETER["Northing at false origin",{y0:.15g},{met},ID["EPSG",8827]]]')
        cf.update(grid_mapping_name='lambert_conformal_conic', standard_parallel=np.array([lat_1, lat_2]) if lat_1 != lat_2 else lat_1,
                  latitude_of_projection_origin=lat_0, longitude_of_central_meridian=lon_0, false_easting=x0, false_northing=y0)
    elif kind == 'stere':
        lat_0 = float(d.get('lat_0', 90.0)); lon_0 = float(d.get('lon_0', 0.0))
        if d.get('lat_ts') is not None:
            conv = (f'CONVERSION["unknown",METHOD["Polar Stereographic (variant B)",ID["EPSG",9829]],PARAMETER["Latitude of standard parallel",{_angle(d["lat_ts"])},{deg},ID["EPSG",8832]],'
                    f'PARAMETER["Longitude of origin",{_angle(lon_0)},{deg},ID["EPSG",8833]],PARAMETER["False easting",{x0:.15g},{met},ID["EPSG",8806]],'
                    f'PARAMETER["False northing",{y0:.15g},{met},ID["EPSG",8807]]]')
            cf.update(standard_parallel=float(d['lat_ts']))
        else:
            conv = (f'CONVERSION["unknown",METHOD["Polar Stereographic (variant A)",ID["EPSG",9810]],PARAMETER["Latitude of natural origin",{_angle(lat_0)},{deg},ID["EPSG",8801]],'
                    f'PARAMETER["Longitude of natural origin",{_angle(lon_0)},{deg},ID["EPSG",8802]],PARAMETER["Scale factor at natural origin",{float(d.get("k_0", 1.0)):.15g},'
                    f'SCALEUNIT["unity",1],ID["EPSG",8805]],PARAMETER["False easting",{x0:.15g},{met},ID["EPSG",8806]],PARAMETER["False northing",{y0:.15g},{met},ID["EPSG",8807]]]')
            cf.update(scale_factor_at_projection_origin=float(d.get('k_0', 1.0)))
        cf.update(grid_mapping_name='polar_stereographic', latitude_of_projection_origin=lat_0, straight_vertical_longitude_from_pole=lon_0, false_easting=x0,
                  false_northing=y0)
    else:
        raise NotImplementedError(f'no CF description of projection {kind!r} here')
    wkt = f'PROJCRS["unknown",{base},{conv},{cs}]'
    cf = dict(crs_wkt=wkt, **cf)
    return wkt, cf

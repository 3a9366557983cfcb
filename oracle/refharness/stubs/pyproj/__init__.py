"""Build-owned stand-in for the subset of `pyproj` that RAiDER's delay hot path touches.

TEST INFRASTRUCTURE ONLY (oracle harness, used only inside the build container to import the
read-only reference from /root/reference and generate tests/golden/*.npz).  Never imported by
the product package `raider_amd`.

pyproj / PROJ are NOT installed in this image, so the geodetic<->ECEF arithmetic below is a
restatement of PROJ's published `cart` conversion (PROJ src/conversions/cart.cpp: `cartesian()`
= Heiskanen&Moritz 5-27, `geodetic()` = Bowring single pass with normalised (c,s) instead of
trig calls).  Parity of THIS arithmetic with the real PROJ is unpinned; the
reference pins only three equatorial ECEF values (test/test_delayFcns.py:86-99) which are
reproduced in tests/test_oracle_geodesy.py.

Supported CRS: EPSG:4326 (lon/lat/h degrees), EPSG:4978 (ECEF metres), and a spherical
Lambert-conformal-conic given as a dict (HRRR grid, models/hrrr.py:248-259).
"""
import numpy as np

from . import exceptions  # noqa: F401

__version__ = '0.0-stub'

# WGS84 (EPSG:7030)
_A = 6378137.0
_RF = 298.257223563
_F = 1.0 / _RF
_ES = 2.0 * _F - _F * _F
_B = (1.0 - _F) * _A
_E2S = _ES / (1.0 - _ES)

DEG_TO_RAD = 0.017453292519943296
RAD_TO_DEG = 57.295779513082321


def _normal_radius(sinphi):
    return _A / np.sqrt(1.0 - _ES * sinphi * sinphi)


def geodetic_to_ecef(lon_deg, lat_deg, h):
    lam = np.asarray(lon_deg, dtype=np.float64) * DEG_TO_RAD
    phi = np.asarray(lat_deg, dtype=np.float64) * DEG_TO_RAD
    h = np.asarray(h, dtype=np.float64)
    cosphi = np.cos(phi)
    sinphi = np.sin(phi)
    N = _normal_radius(sinphi)
    x = (N + h) * cosphi * np.cos(lam)
    y = (N + h) * cosphi * np.sin(lam)
    z = (N * (1.0 - _ES) + h) * sinphi
    return x, y, z


def ecef_to_geodetic(x, y, z):
    x = np.asarray(x, dtype=np.float64)
    y = np.asarray(y, dtype=np.float64)
    z = np.asarray(z, dtype=np.float64)
    p = np.hypot(x, y)
    y_theta = z * _A
    x_theta = p * _B
    norm = np.hypot(y_theta, x_theta)
    with np.errstate(invalid='ignore', divide='ignore'):
        c = np.where(norm == 0, 1.0, x_theta / norm)
        s = np.where(norm == 0, 0.0, y_theta / norm)
        y_phi = z + _E2S * _B * s * s * s
        x_phi = p - _ES * _A * c * c * c
        norm_phi = np.hypot(y_phi, x_phi)
        cosphi = np.where(norm_phi == 0, 1.0, x_phi / norm_phi)
        sinphi = np.where(norm_phi == 0, 0.0, y_phi / norm_phi)
        phi = np.arctan(y_phi / x_phi)
        polar = x_phi <= 0
        phi = np.where(polar, np.where(z >= 0, np.pi / 2, -np.pi / 2), phi)
        cosphi = np.where(polar, 0.0, cosphi)
        sinphi = np.where(polar, np.where(z >= 0, 1.0, -1.0), sinphi)
        lam = np.arctan2(y, x)
        h_reg = p / cosphi - _normal_radius(sinphi)
        # geocentric radius branch poleward of 89.99994 deg
        r = np.hypot(_A * _A * cosphi, _B * _B * sinphi) / np.hypot(_A * cosphi, _B * sinphi)
        h = np.where(cosphi < 1e-6, np.abs(z) - r, h_reg)
    return lam * RAD_TO_DEG, phi * RAD_TO_DEG, h


class _Axis:
    def __init__(self, unit_name, direction):
        self.unit_name = unit_name
        self.direction = direction


class CRS:
    """Minimal CRS: compares by a normalised key."""

    def __init__(self, spec=4326):
        if isinstance(spec, CRS):
            self._key = spec._key
            self._params = spec._params
            return
        self._params = None
        if isinstance(spec, dict):
            self._key = ('lcc', tuple(sorted(spec.items())))
            self._params = dict(spec)
            return
        if isinstance(spec, str):
            s = spec.strip()
            if s.upper().startswith('EPSG:'):
                s = s.split(':')[-1]
            if s.startswith('STUBWKT:'):
                s = s.split(':')[-1]
            try:
                spec = int(s)
            except ValueError:
                raise exceptions.CRSError(f'stub pyproj cannot parse CRS {spec!r}')
        if int(spec) not in (4326, 4978):
            raise exceptions.CRSError(f'stub pyproj supports EPSG 4326/4978 only, got {spec}')
        self._key = ('epsg', int(spec))

    @classmethod
    def from_epsg(cls, code):
        return cls(code)

    @classmethod
    def from_wkt(cls, wkt):
        return cls(wkt)

    @classmethod
    def from_user_input(cls, x):
        return cls(x)

    def to_epsg(self):
        return self._key[1] if self._key[0] == 'epsg' else None

    def to_wkt(self):
        return f'STUBWKT:{self._key[1]}'

    def to_cf(self):
        return {'crs_wkt': self.to_wkt(), 'grid_mapping_name': 'latitude_longitude'}

    @property
    def axis_info(self):
        if self._key == ('epsg', 4326):
            return [_Axis('degree', 'north'), _Axis('degree', 'east')]
        return [_Axis('metre', 'east'), _Axis('metre', 'north')]

    def __eq__(self, other):
        try:
            other = other if isinstance(other, CRS) else CRS(other)
        except Exception:
            return False
        return self._key == other._key

    def __ne__(self, other):
        return not self.__eq__(other)

    def __hash__(self):
        return hash(self._key)

    def __repr__(self):
        return f'<stub CRS {self._key}>'


def _lcc_params(p):
    R = float(p['R'])
    lat1 = np.deg2rad(p['lat_1'])
    lat2 = np.deg2rad(p['lat_2'])
    lat0 = np.deg2rad(p['lat_0'])
    lon0 = np.deg2rad(p['lon_0'])
    if abs(lat1 - lat2) < 1e-10:
        n = np.sin(lat1)
    else:
        n = np.log(np.cos(lat1) / np.cos(lat2)) / np.log(
            np.tan(np.pi / 4 + lat2 / 2) / np.tan(np.pi / 4 + lat1 / 2))
    F = np.cos(lat1) * np.tan(np.pi / 4 + lat1 / 2) ** n / n
    rho0 = R * F / np.tan(np.pi / 4 + lat0 / 2) ** n
    return R, n, F, rho0, lon0


class Transformer:
    def __init__(self, src, dst):
        self._src = CRS(src)
        self._dst = CRS(dst)

    @classmethod
    def from_crs(cls, src, dst, always_xy=False):
        if not always_xy:
            raise NotImplementedError('stub pyproj Transformer needs always_xy=True')
        return cls(src, dst)

    def _to_llh(self, crs, x, y, z):
        if crs._key == ('epsg', 4326):
            return x, y, z
        if crs._key == ('epsg', 4978):
            return ecef_to_geodetic(x, y, z)
        R, n, F, rho0, lon0 = _lcc_params(crs._params)
        x = np.asarray(x, float); y = np.asarray(y, float)
        rho = np.sign(n) * np.hypot(x, rho0 - y)
        theta = np.arctan2(x, rho0 - y)
        lat = 2 * np.arctan((R * F / rho) ** (1 / n)) - np.pi / 2
        lon = lon0 + theta / n
        return np.rad2deg(lon), np.rad2deg(lat), z

    def _from_llh(self, crs, lon, lat, h):
        if crs._key == ('epsg', 4326):
            return lon, lat, h
        if crs._key == ('epsg', 4978):
            return geodetic_to_ecef(lon, lat, h)
        R, n, F, rho0, lon0 = _lcc_params(crs._params)
        lam = np.deg2rad(np.asarray(lon, float)); phi = np.deg2rad(np.asarray(lat, float))
        dlam = lam - lon0
        dlam = (dlam + np.pi) % (2 * np.pi) - np.pi
        rho = R * F / np.tan(np.pi / 4 + phi / 2) ** n
        return rho * np.sin(n * dlam), rho0 - rho * np.cos(n * dlam), h

    def transform(self, xx, yy, zz=None, **kw):
        if zz is None:
            zz = np.zeros_like(np.asarray(xx, dtype=float))
        if self._src == self._dst:
            return xx, yy, zz
        lon, lat, h = self._to_llh(self._src, xx, yy, zz)
        return self._from_llh(self._dst, lon, lat, h)


class Proj:
    def __init__(self, *a, **k):
        raise NotImplementedError('stub pyproj.Proj')

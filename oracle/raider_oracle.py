"""CPU oracle: NumPy restatement of RAiDER's tropospheric-delay hot path.

*** TEST INFRASTRUCTURE - NOT PART OF THE PRODUCT. ***
Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import this module,
and only as the checker / reported baseline.  The product package `raider_amd` never imports it and
fails loudly when its HIP library is missing.

Parity status
-------------
Pinned against the reference itself: `oracle/refharness/gen_golden.py` imports the unmodified
reference from /root/reference (in the build container), runs its `_build_cube`,
`_build_cube_ray`, `build_ray`, `getTopOfAtmosphere`, `inc_hd_to_enu`, `enu2ecef`,
`getZenithLookVecs`, `Conventional.__call__` arithmetic, the native `interpolate`,
`interpolate_along_axis` and `makePoints*D`, the `WeatherModel` processing methods, the azimuth-time
weighting functions and the orbit-file readers on seeded inputs and commits inputs+outputs under
tests/golden/*.npz; `tests/test_oracle_golden.py` checks every function below against them, and
`tests/test_oracle_vs_reference.py` re-runs the comparison LIVE on fresh random cases wherever the
reference tree is present.  Data files of the reference's own test suite pin the rest: processed
ERA-5 cubes written by the real RAiDER, the raw model-level files they were made from, its GMAO
time-interpolation product (tests/golden/ref_files, tests/test_ref_files.py).
UNPINNED against the reference's third-party dependencies (stated in DESIGN.md 6): pyproj/PROJ is
not installed here and not under /root/reference, so `ecef2lla`/`lla2ecef` restate PROJ's published
`cart` conversion and `lcc_forward` its `lcc` projection; both ARE pinned on authorities independent
of PROJ (IOGP Guidance Note 7-2's worked example; Snyder's USGS PP 1395 numerical examples) and on
the three exact ECEF values of `test/test_delayFcns.py:86-99`, but not on PROJ's own binary.
isce3's geo2rdr (orbit -> look vectors) is absent too: `orbit_look_vectors` restates the zero-Doppler
solve from its call sites and is pinned by geometry only.

Every function cites the reference file:line (relative to /root/reference/) it follows.
"""
import numpy as np

# ----------------------------------------------------------------------------------------------
# constants  (tools/RAiDER/constants.py:12-23)
# ----------------------------------------------------------------------------------------------
_ZMIN = np.float64(-100)
_ZREF = np.float64(26000)
_STEP = np.float64(15.0)
R_EARTH_MAX_WGS84 = 6378137
R_EARTH_MIN_WGS84 = 6356752
_CUBE_SPACING_IN_M = float(2000)

# WGS84 ellipsoid as PROJ derives it from (a, rf)
WGS84_A = 6378137.0
WGS84_RF = 298.257223563
WGS84_F = 1.0 / WGS84_RF
WGS84_ES = 2.0 * WGS84_F - WGS84_F * WGS84_F          # first eccentricity squared
WGS84_B = (1.0 - WGS84_F) * WGS84_A
WGS84_E2S = WGS84_ES / (1.0 - WGS84_ES)               # second eccentricity squared
DEG_TO_RAD = 0.017453292519943296
RAD_TO_DEG = 57.295779513082321


# ----------------------------------------------------------------------------------------------
# geodesy
# ----------------------------------------------------------------------------------------------
def cosd(x):
    """utilFcns.py:72-74"""
    return np.cos(np.radians(x))


def sind(x):
    """utilFcns.py:67-69"""
    return np.sin(np.radians(x))


def lla2ecef(lat, lon, height):
    """utilFcns.py:77-81 -> pyproj Transformer(4326->4978, always_xy) -> PROJ cart `cartesian()`.

    Returns (x, y, z)."""
    lam = np.asarray(lon, dtype=np.float64) * DEG_TO_RAD
    phi = np.asarray(lat, dtype=np.float64) * DEG_TO_RAD
    h = np.asarray(height, dtype=np.float64)
    cosphi = np.cos(phi)
    sinphi = np.sin(phi)
    N = WGS84_A / np.sqrt(1.0 - WGS84_ES * sinphi * sinphi)
    x = (N + h) * cosphi * np.cos(lam)
    y = (N + h) * cosphi * np.sin(lam)
    z = (N * (1.0 - WGS84_ES) + h) * sinphi
    return x, y, z


def ecef2lla(x, y, z):
    """utilFcns.py:84-88 -> pyproj Transformer(4978->4326, always_xy) -> PROJ cart `geodetic()`.

    Returns (lon_deg, lat_deg, h)  (always_xy order, as the reference's callers index it:
    `pos_llh[2]` is the height, losreader.py:730-731)."""
    x = np.asarray(x, dtype=np.float64)
    y = np.asarray(y, dtype=np.float64)
    z = np.asarray(z, dtype=np.float64)
    p = np.hypot(x, y)
    y_theta = z * WGS84_A
    x_theta = p * WGS84_B
    norm = np.hypot(y_theta, x_theta)
    with np.errstate(invalid='ignore', divide='ignore'):
        c = np.where(norm == 0, 1.0, x_theta / norm)
        s = np.where(norm == 0, 0.0, y_theta / norm)
        y_phi = z + WGS84_E2S * WGS84_B * s * s * s
        x_phi = p - WGS84_ES * WGS84_A * c * c * c
        norm_phi = np.hypot(y_phi, x_phi)
        cosphi = np.where(norm_phi == 0, 1.0, x_phi / norm_phi)
        sinphi = np.where(norm_phi == 0, 0.0, y_phi / norm_phi)
        phi = np.arctan(y_phi / x_phi)
        polar = x_phi <= 0
        phi = np.where(polar, np.where(z >= 0, np.pi / 2, -np.pi / 2), phi)
        cosphi = np.where(polar, 0.0, cosphi)
        sinphi = np.where(polar, np.where(z >= 0, 1.0, -1.0), sinphi)
        lam = np.arctan2(y, x)
        h_reg = p / cosphi - WGS84_A / np.sqrt(1.0 - WGS84_ES * sinphi * sinphi)
        r = np.hypot(WGS84_A * WGS84_A * cosphi, WGS84_B * WGS84_B * sinphi) / np.hypot(
            WGS84_A * cosphi, WGS84_B * sinphi)
        h = np.where(cosphi < 1e-6, np.abs(z) - r, h_reg)
    return lam * RAD_TO_DEG, phi * RAD_TO_DEG, h


def enu2ecef(east, north, up, lat0, lon0, h0):
    """utilFcns.py:91-121 (h0 unused there too). Returns (...,3)."""
    t = cosd(lat0) * up - sind(lat0) * north
    w = sind(lat0) * up + cosd(lat0) * north
    u = cosd(lon0) * t - sind(lon0) * east
    v = sind(lon0) * t + cosd(lon0) * east
    return np.stack((u, v, w), axis=-1)


def ecef2enu(xyz, lat, lon, height):
    """utilFcns.py:124-137."""
    x, y, z = xyz[..., 0], xyz[..., 1], xyz[..., 2]
    t = cosd(lon) * x + sind(lon) * y
    e = -sind(lon) * x + cosd(lon) * y
    n = -sind(lat) * t + cosd(lat) * z
    u = cosd(lat) * t + sind(lat) * z
    return np.stack((e, n, u), axis=-1)


def inc_hd_to_enu(incidence, heading):
    """losreader.py:374-396."""
    if np.any(np.asarray(incidence) < 0):
        raise ValueError('inc_hd_to_enu: Incidence angle cannot be less than 0')
    east = sind(incidence) * cosd(heading + 90)
    north = sind(incidence) * sind(heading + 90)
    up = cosd(incidence)
    return np.stack((east, north, up), axis=-1)


def getZenithLookVecs(lats, lons, heights):
    """losreader.py:302-316."""
    x = np.cos(np.radians(lats)) * np.cos(np.radians(lons))
    y = np.cos(np.radians(lats)) * np.sin(np.radians(lons))
    z = np.sin(np.radians(lats))
    return np.stack([x, y, z], axis=-1)


def conventional_project(delays, LOS_enu):
    """losreader.py:130-133 (`Conventional.__call__` tail)."""
    if delays.shape == LOS_enu.shape:
        return delays / LOS_enu
    return delays / LOS_enu[..., -1]


def look_vectors_from_inc_hd(inc, hd, lat, lon, ht):
    """The duck-typed LOS used for goldens/bench: inc/heading -> ENU -> ECEF.

    losreader.py:374-396 + utilFcns.py:91-121."""
    enu = inc_hd_to_enu(inc, hd)
    return enu2ecef(enu[..., 0], enu[..., 1], enu[..., 2], lat, lon, ht)


def lcc_forward(lat, lon, lat_1, lat_2, lat_0, lon_0, x_0=0.0, y_0=0.0, a=6371229.0, es=0.0):
    """Geodetic -> Lambert conformal conic, what pyproj does for a projected model CRS (HRRR: models/hrrr.py:248-259;
    call sites delay.py:207-209,253,295).  Restates PROJ's published `lcc` (Snyder 15-1..15-4); PARITY WITH PROJ ITSELF
    IS UNPINNED (pyproj is not installed here).  Returns (x, y)."""
    e = np.sqrt(es)

    def tsfn(phi):
        s_ = np.sin(phi)
        t_ = np.tan(0.5 * (np.pi / 2 - phi))
        return t_ / ((1 - e * s_) / (1 + e * s_)) ** (0.5 * e) if e != 0 else t_

    def msfn(phi):
        return np.cos(phi) / np.sqrt(1 - es * np.sin(phi) ** 2)
    p1, p2, p0 = np.radians(lat_1), np.radians(lat_2), np.radians(lat_0)
    if abs(p1 - p2) >= 1e-10:
        n = np.log(msfn(p1) / msfn(p2)) / np.log(tsfn(p1) / tsfn(p2))
    else:
        n = np.sin(p1)
    F = msfn(p1) * tsfn(p1) ** (-n) / n
    rho0 = a * F * tsfn(p0) ** n
    dlam = np.radians(np.asarray(lon, dtype=np.float64)) - np.radians(lon_0)
    dlam = np.where(dlam > np.pi, dlam - 2 * np.pi, np.where(dlam < -np.pi, dlam + 2 * np.pi, dlam))
    rho = a * F * tsfn(np.radians(np.asarray(lat, dtype=np.float64))) ** n
    return x_0 + rho * np.sin(n * dlam), y_0 + rho0 - rho * np.cos(n * dlam)


# ----------------------------------------------------------------------------------------------
# scipy RegularGridInterpolator (linear, bounds_error=False, fill_value=nan) restated
#   call sites: delayFcns.py:55-56, delay.py:214,319,120-121
#   algorithm: scipy/interpolate/_rgi.py:405-443,470-499,585-592 + _rgi_cython.find_indices (v1.15.3)
# ----------------------------------------------------------------------------------------------
class RGI:
    """Rectilinear N-D (N<=3 used here) linear interpolator with scipy semantics.

    `.grid` is exposed because delay.py:239 reads `interpolators[0].grid[2]`."""

    def __init__(self, points, values, fill_value=np.nan):
        grid = [np.asarray(p, dtype=np.float64) for p in points]
        values = np.asarray(values)
        # scipy flips descending axes (scipy _rgi.py `_check_points`)
        for i, g in enumerate(grid):
            if g.size > 1 and np.all(np.diff(g) < 0):
                grid[i] = g[::-1].copy()
                values = np.flip(values, axis=i)
            elif g.size > 1 and not np.all(np.diff(g) > 0):
                raise ValueError(f'The points in dimension {i} must be strictly ascending or descending')
        self.grid = tuple(grid)
        self.values = values
        self.fill_value = fill_value

    def __call__(self, xi):
        xi = np.asarray(xi, dtype=np.float64)
        nd = len(self.grid)
        shp = xi.shape
        xi = xi.reshape(-1, nd)
        nans = np.any(np.isnan(xi), axis=-1)
        oob = np.zeros(xi.shape[0], dtype=bool)
        idx, t = [], []
        for d, g in enumerate(self.grid):
            x = xi[:, d]
            oob |= x < g[0]
            oob |= x > g[-1]
            i = np.clip(np.searchsorted(g, x, side='right') - 1, 0, g.size - 2)
            with np.errstate(invalid='ignore', divide='ignore'):
                td = (x - g[i]) / (g[i + 1] - g[i])
            idx.append(i)
            t.append(td)
        value = np.zeros(xi.shape[0], dtype=np.float64)
        # hypercube corners in lexicographic order, last axis fastest (_rgi.py:490-498)
        for corner in range(1 << nd):
            bits = [(corner >> (nd - 1 - d)) & 1 for d in range(nd)]
            weight = np.ones(xi.shape[0])
            for d in range(nd):
                weight = weight * (t[d] if bits[d] else (1 - t[d]))
            edge = tuple(idx[d] + bits[d] for d in range(nd))
            value = value + np.asarray(self.values[edge]) * weight
        value[oob] = self.fill_value
        value[nans] = np.nan
        return value.reshape(shp[:-1])


def getInterpolators(xs, ys, zs, wet_zyx, hydro_zyx):
    """delayFcns.py:23-58 with the file read factored out: fields arrive in file order (z,y,x)
    and are transposed to (y,x,z) (delayFcns.py:40-41)."""
    wet = np.asarray(wet_zyx).transpose(1, 2, 0)
    hydro = np.asarray(hydro_zyx).transpose(1, 2, 0)
    return RGI((ys, xs, zs), wet), RGI((ys, xs, zs), hydro)


# ----------------------------------------------------------------------------------------------
# ray geometry
# ----------------------------------------------------------------------------------------------
def getTopOfAtmosphere(xyz, look_vecs, toaheight, factor=None):
    """losreader.py:706-733."""
    if factor is not None:
        maxIter = 3
    else:
        maxIter = 10
        factor = 1.0
    pos = xyz + toaheight * look_vecs
    for _ in range(maxIter):
        pos_llh = ecef2lla(pos[..., 0], pos[..., 1], pos[..., 2])
        pos = pos + look_vecs * ((toaheight - pos_llh[2]) / factor)[..., None]
    return pos


def ray_levels(model_zs, ht, zref):
    """The slice-uniform part of build_ray (losreader.py:785-808): which model intervals
    contribute and their clipped [low_ht, high_ht].  Returns list of (low_ht, high_ht)."""
    model_zs = np.asarray(model_zs, dtype=np.float64)
    out = []
    for zz in range(model_zs.size - 1):
        low_ht = model_zs[zz]
        high_ht = model_zs[zz + 1]
        if high_ht == model_zs[-1]:
            high_ht = high_ht - 0.01
        if (high_ht < ht) or (low_ht >= zref):
            continue
        if low_ht < ht:
            low_ht = ht
        if high_ht > zref:
            high_ht = zref
        if np.abs(high_ht - low_ht) < 1.0:
            continue
        out.append((float(low_ht), float(high_ht)))
    return out


def build_ray(model_zs, ht, xyz, LOS, MAX_TROPO_HEIGHT=_ZREF):
    """losreader.py:772-835."""
    low_xyz = None
    high_xyz = None
    cos_factor = None
    ray_lengths, low_xyzs, high_xyzs = [], [], []
    for low_ht, high_ht in ray_levels(model_zs, ht, MAX_TROPO_HEIGHT):
        if high_xyz is not None:
            low_xyz = high_xyz
        else:
            low_xyz = getTopOfAtmosphere(xyz, LOS, low_ht, factor=cos_factor)
        high_xyz = getTopOfAtmosphere(xyz, LOS, high_ht, factor=cos_factor)
        ray_length = np.linalg.norm(high_xyz - low_xyz, axis=-1)
        if cos_factor is None:
            cos_factor = (high_ht - low_ht) / ray_length
        ray_lengths.append(ray_length)
        low_xyzs.append(low_xyz)
        high_xyzs.append(high_xyz)
    if not ray_lengths:
        return None, None, None
    return np.stack(ray_lengths), np.stack(low_xyzs), np.stack(high_xyzs)


def nparts_from_lengths(ray_lengths, MAX_SEGMENT_LENGTH=1000.0):
    """delay.py:283 (max over the WHOLE slice; NaN poisons, as ndarray.max does)."""
    K = ray_lengths.shape[0]
    return np.ceil(ray_lengths.reshape(K, -1).max(1) / MAX_SEGMENT_LENGTH).astype(int) + 1


# ----------------------------------------------------------------------------------------------
# look vectors from orbit state vectors  (Raytracing.getLookVectors, losreader.py:219-255)
#   The reference calls isce3 (geometry.geo2rdr with an empty Doppler LUT + Orbit.interpolate); isce3 is a third-party
#   dependency that is neither installed here nor under /root/reference (environment.yml:25 `isce3>=0.15.0`).
#   Restated from its published algorithm: zero-Doppler Newton on azimuth time with 4-point Hermite orbit
#   interpolation.  PARITY WITH isce3 IS UNPINNED; only geometric properties are tested.
# ----------------------------------------------------------------------------------------------
def orbit_hermite(st, sp, sv, t):
    """4-point Hermite interpolation of position and velocity at times t (array) from state vectors
    st[n] (s), sp[n,3], sv[n,3]."""
    t = np.atleast_1d(np.asarray(t, dtype=np.float64))
    n = st.size
    lo = np.searchsorted(st, t, side='right')
    i0 = np.clip(lo - 2, 0, n - 4)
    tt = np.stack([st[i0 + k] for k in range(4)], -1)            # (m, 4)
    X = np.stack([sp[i0 + k] for k in range(4)], 1)               # (m, 4, 3)
    V = np.stack([sv[i0 + k] for k in range(4)], 1)
    f1 = t[:, None] - tt
    pos = np.zeros((t.size, 3)); vel = np.zeros((t.size, 3))
    for i in range(4):
        others = [j for j in range(4) if j != i]
        ssum = sum(1.0 / (tt[:, i] - tt[:, j]) for j in others)
        f0 = 1.0 - 2.0 * (t - tt[:, i]) * ssum
        h = np.ones(t.size)
        for k in others:
            h = h * (t - tt[:, k]) / (tt[:, i] - tt[:, k])
        hdot = np.zeros(t.size)
        for j in others:
            p2 = np.ones(t.size)
            for k in others:
                if k != j:
                    p2 = p2 * (t - tt[:, k]) / (tt[:, i] - tt[:, k])
            hdot = hdot + p2 / (tt[:, i] - tt[:, j])
        g1 = h + 2.0 * (t - tt[:, i]) * hdot
        g0 = 2.0 * (f0 * hdot - h * ssum)
        pos += (X[:, i] * f0[:, None] + V[:, i] * f1[:, i][:, None]) * (h * h)[:, None]
        vel += (X[:, i] * g0[:, None] + V[:, i] * g1[:, None]) * h[:, None]
    return pos, vel


def orbit_look_vectors(st, sp, sv, xyz, threshold=1.0e-7, maxiter=30):
    """Zero-Doppler geo2rdr per target + unit look vector target->sensor.  Returns (los[...,3], aztime, slant_range)."""
    shp = xyz.shape[:-1]
    T = np.asarray(xyz, dtype=np.float64).reshape(-1, 3)
    t = np.full(T.shape[0], 0.5 * (st[0] + st[-1]))
    done = np.zeros(T.shape[0], dtype=bool)
    for _ in range(maxiter):
        pos, vel = orbit_hermite(st, sp, sv, t)
        d = T - pos
        fn = np.sum(d * vel, -1)
        fnp = -np.sum(vel * vel, -1)
        step = np.where(done, 0.0, fn / fnp)
        t = t - step
        done |= np.abs(step) < threshold
        if done.all():
            break
    pos, _ = orbit_hermite(st, sp, sv, np.where(np.isnan(t), st[0], t))
    d = pos - T
    rg = np.linalg.norm(d, axis=-1)
    bad = ~done | (t < st[0]) | (t > st[-1]) | np.isnan(T).any(-1)
    los = d / rg[:, None]
    los[bad] = np.nan; rg = np.where(bad, np.nan, rg); t = np.where(bad, np.nan, t)
    return los.reshape(shp + (3,)), t.reshape(shp), rg.reshape(shp)


# ----------------------------------------------------------------------------------------------
# cube builders
# ----------------------------------------------------------------------------------------------
def stere_forward(lat, lon, lat_0=90.0, lat_ts=None, k_0=1.0, lon_0=0.0, x_0=0.0, y_0=0.0, a=6371229.0, es=0.0):
    """Geodetic -> POLAR stereographic (lat_0 = +-90), what pyproj does for a model on `+proj=stere` (HRRR-AK:
    models/hrrr.py:22-25,359 `+proj=stere +lat_0=90 +lon_0=225 +lat_ts=60` on the sphere a = b = 6371229; call sites
    delay.py:207-209,253,295).  Restates PROJ's published `stere` polar branch = Snyder (USGS PP 1395) 21-33 / 21-34 with
    21-39 (true scale at lat_ts) or 21-33 (scale k_0 at the pole); the southern aspect mirrors latitude, as PROJ does.
    PARITY WITH PROJ ITSELF IS UNPINNED (pyproj is not installed here); pinned on Snyder's numerical example (p. 315:
    International ellipsoid, lat_ts = -71, lon_0 = -100, point (-75, 150) -> x = -1 540 033.6 m, y = -560 526.4 m).
    Returns (x, y)."""
    if abs(abs(lat_0) - 90.0) > 1e-9:
        raise NotImplementedError('only the polar aspect of the stereographic projection (lat_0 = +-90)')
    e = np.sqrt(es)
    south = lat_0 < 0
    sg = -1.0 if south else 1.0

    def tsfn(phi):
        s_ = np.sin(phi)
        t_ = np.tan(0.5 * (np.pi / 2 - phi))
        return t_ / ((1 - e * s_) / (1 + e * s_)) ** (0.5 * e) if e != 0 else t_
    phi = sg * np.radians(np.asarray(lat, dtype=np.float64))
    t = tsfn(phi)
    if lat_ts is not None and abs(abs(lat_ts) - 90.0) > 1e-9:
        pc = abs(np.radians(lat_ts))
        mc = np.cos(pc) / np.sqrt(1 - es * np.sin(pc) ** 2)
        rho = a * mc * t / tsfn(pc)
    else:
        rho = 2 * a * k_0 * t / np.sqrt((1 + e) ** (1 + e) * (1 - e) ** (1 - e))
    dlam = np.radians(np.asarray(lon, dtype=np.float64)) - np.radians(lon_0)
    dlam = np.where(dlam > np.pi, dlam - 2 * np.pi, np.where(dlam < -np.pi, dlam + 2 * np.pi, dlam))
    return x_0 + rho * np.sin(dlam), y_0 - sg * rho * np.cos(dlam)


def _tm_setup(a, es, lat_0, lon_0, k_0):
    f = 1 - np.sqrt(1 - es); n = f / (2 - f)
    A = a / (1 + n) * (1 + n ** 2 / 4 + n ** 4 / 64 + n ** 6 / 256)
    al = [n / 2 - 2 * n ** 2 / 3 + 5 * n ** 3 / 16 + 41 * n ** 4 / 180 - 127 * n ** 5 / 288 + 7891 * n ** 6 / 37800,
          13 * n ** 2 / 48 - 3 * n ** 3 / 5 + 557 * n ** 4 / 1440 + 281 * n ** 5 / 630 - 1983433 * n ** 6 / 1935360,
          61 * n ** 3 / 240 - 103 * n ** 4 / 140 + 15061 * n ** 5 / 26880 + 167603 * n ** 6 / 181440,
          49561 * n ** 4 / 161280 - 179 * n ** 5 / 168 + 6601661 * n ** 6 / 7257600,
          34729 * n ** 5 / 80640 - 3418889 * n ** 6 / 1995840, 212378941 * n ** 6 / 319334400]
    be = [n / 2 - 2 * n ** 2 / 3 + 37 * n ** 3 / 96 - n ** 4 / 360 - 81 * n ** 5 / 512 + 96199 * n ** 6 / 604800,
          n ** 2 / 48 + n ** 3 / 15 - 437 * n ** 4 / 1440 + 46 * n ** 5 / 105 - 1118711 * n ** 6 / 3870720,
          17 * n ** 3 / 480 - 37 * n ** 4 / 840 - 209 * n ** 5 / 4480 + 5569 * n ** 6 / 90720,
          4397 * n ** 4 / 161280 - 11 * n ** 5 / 504 - 830251 * n ** 6 / 7257600,
          4583 * n ** 5 / 161280 - 108847 * n ** 6 / 3991680, 20648693 * n ** 6 / 638668800]
    return np.sqrt(es), k_0 * A, al, be


def _tm_conformal(e, phi, lam):
    tau = np.tan(phi)
    sig = np.sinh(e * np.arctanh(e * tau / np.sqrt(1 + tau ** 2)))
    taup = tau * np.sqrt(1 + sig ** 2) - sig * np.sqrt(1 + tau ** 2)
    return np.arctan2(taup, np.cos(lam)), np.arcsinh(np.sin(lam) / np.sqrt(taup ** 2 + np.cos(lam) ** 2))


def tm_forward(lat, lon, lat_0=0.0, lon_0=0.0, k_0=0.9996, x_0=500000.0, y_0=0.0, a=6378137.0, es=0.0066943799901413165):
    """Geodetic -> transverse Mercator (UTM: lat_0 = 0, k_0 = 0.9996, x_0 = 500 000, y_0 = 0 / 10 000 000, lon_0 = 6 zone - 183):
    what pyproj does in transformPoints (delay.py:404-436) for a UTM output grid.  Krueger's series in the third flattening to
    n^6 (Karney 2011, eqs. 7-11, 35) = the formulation of PROJ's `etmerc` / `utm`; PARITY WITH PROJ ITSELF IS UNPINNED (pyproj is
    absent), pinned instead on Snyder's UTM example (PP 1395 p. 269: Clarke 1866, (40.5 N, 73.5 W), lon_0 = 75 W -> x = 127 106.5 m,
    y = 4 484 124.4 m) and the OSGB example of IOGP Guidance Note 7-2 (E 577 274.99, N 69 740.50).  Returns (x, y)."""
    e, kA, al, _ = _tm_setup(a, es, lat_0, lon_0, k_0)
    xi0p, eta0p = _tm_conformal(e, np.radians(lat_0), 0.0)
    xi0 = xi0p + sum(al[j] * np.sin(2 * (j + 1) * xi0p) * np.cosh(2 * (j + 1) * eta0p) for j in range(6))
    lam = np.radians(np.asarray(lon, dtype=np.float64) - lon_0)
    lam = np.where(lam > np.pi, lam - 2 * np.pi, np.where(lam < -np.pi, lam + 2 * np.pi, lam))
    xip, etap = _tm_conformal(e, np.radians(np.asarray(lat, dtype=np.float64)), lam)
    xi = xip + sum(al[j] * np.sin(2 * (j + 1) * xip) * np.cosh(2 * (j + 1) * etap) for j in range(6))
    eta = etap + sum(al[j] * np.cos(2 * (j + 1) * xip) * np.sinh(2 * (j + 1) * etap) for j in range(6))
    return x_0 + kA * eta, y_0 + kA * (xi - xi0)


def tm_inverse(x, y, lat_0=0.0, lon_0=0.0, k_0=0.9996, x_0=500000.0, y_0=0.0, a=6378137.0, es=0.0066943799901413165):
    """Transverse Mercator -> geodetic (Karney 2011 eqs. 11, 36 and the Newton step 19-21).  Returns (lat, lon) in degrees."""
    e, kA, al, be = _tm_setup(a, es, lat_0, lon_0, k_0)
    xi0p, eta0p = _tm_conformal(e, np.radians(lat_0), 0.0)
    xi0 = xi0p + sum(al[j] * np.sin(2 * (j + 1) * xi0p) * np.cosh(2 * (j + 1) * eta0p) for j in range(6))
    xi = (np.asarray(y, dtype=np.float64) - y_0) / kA + xi0; eta = (np.asarray(x, dtype=np.float64) - x_0) / kA
    xip = xi - sum(be[j] * np.sin(2 * (j + 1) * xi) * np.cosh(2 * (j + 1) * eta) for j in range(6))
    etap = eta - sum(be[j] * np.cos(2 * (j + 1) * xi) * np.sinh(2 * (j + 1) * eta) for j in range(6))
    taup = np.sin(xip) / np.sqrt(np.sinh(etap) ** 2 + np.cos(xip) ** 2)
    lam = np.arctan2(np.sinh(etap), np.cos(xip))
    tau = taup.copy() if isinstance(taup, np.ndarray) else np.float64(taup)
    for _ in range(6):
        t1 = np.sqrt(1 + tau ** 2)
        sig = np.sinh(e * np.arctanh(e * tau / t1))
        tpi = tau * np.sqrt(1 + sig ** 2) - sig * t1
        tau = tau + (taup - tpi) / np.sqrt(1 + tpi ** 2) * (1 + (1 - es) * tau ** 2) / ((1 - es) * t1)
    lon = np.degrees(lam) + lon_0
    lon = np.where(lon > 180, lon - 360, np.where(lon < -180, lon + 360, lon))
    return np.degrees(np.arctan(tau)), lon


def _phi_from_t(t, e):
    """Latitude from the conformal quantity t (Snyder 7-9 / 15-9; PROJ pj_phi2): closed form on the sphere, fixed-point on the ellipsoid."""
    phi = np.pi / 2 - 2 * np.arctan(t)
    if e != 0:
        for _ in range(15):
            es_ = e * np.sin(phi)
            phi = np.pi / 2 - 2 * np.arctan(t * ((1 - es_) / (1 + es_)) ** (0.5 * e))
    return phi


def lcc_inverse(x, y, lat_1, lat_2, lat_0, lon_0, x_0=0.0, y_0=0.0, a=6371229.0, es=0.0):
    """Lambert conformal conic -> geodetic (Snyder 15-10, 15-11, 14-9, 15-9; PROJ `lcc` inverse) - the way back of
    transformPoints(..., hrrr_proj, 4326) (test/test_delayFcns.py:67-84 round-trips it).  Pinned on Snyder's inverse examples
    (pp. 296-298).  Returns (lat, lon) in degrees."""
    e = np.sqrt(es)

    def tsfn(phi):
        s_ = np.sin(phi)
        t_ = np.tan(0.5 * (np.pi / 2 - phi))
        return t_ / ((1 - e * s_) / (1 + e * s_)) ** (0.5 * e) if e != 0 else t_

    def msfn(phi):
        return np.cos(phi) / np.sqrt(1 - es * np.sin(phi) ** 2)
    p1, p2, p0 = np.radians(lat_1), np.radians(lat_2), np.radians(lat_0)
    n = np.log(msfn(p1) / msfn(p2)) / np.log(tsfn(p1) / tsfn(p2)) if abs(p1 - p2) >= 1e-10 else np.sin(p1)
    F = msfn(p1) * tsfn(p1) ** (-n) / n
    rho0 = a * F * tsfn(p0) ** n
    sg = np.sign(n)                                              # Snyder: rho takes the sign of n (and so do x, rho0 - y in 14-11)
    xp = sg * (np.asarray(x, dtype=np.float64) - x_0); yp = sg * (rho0 - (np.asarray(y, dtype=np.float64) - y_0))
    rho = sg * np.hypot(xp, yp)
    with np.errstate(divide='ignore', invalid='ignore'):
        t = (rho / (a * F)) ** (1.0 / n)
    lat = np.degrees(_phi_from_t(t, e))
    lon = np.degrees(np.radians(lon_0) + np.arctan2(xp, yp) / n)
    lat = np.where(rho == 0, 90.0 * sg, lat); lon = np.where(rho == 0, lon_0, lon)
    lon = np.where(lon > 180, lon - 360, np.where(lon < -180, lon + 360, lon))
    return lat, lon


def stere_inverse(x, y, lat_0=90.0, lat_ts=None, k_0=1.0, lon_0=0.0, x_0=0.0, y_0=0.0, a=6371229.0, es=0.0):
    """POLAR stereographic -> geodetic (Snyder 20-18, 21-39 / 21-40, 7-9; PROJ `stere` polar inverse).  Pinned on Snyder's inverse
    example (p. 317).  Returns (lat, lon) in degrees."""
    if abs(abs(lat_0) - 90.0) > 1e-9:
        raise NotImplementedError('only the polar aspect of the stereographic projection (lat_0 = +-90)')
    e = np.sqrt(es)
    sg = -1.0 if lat_0 < 0 else 1.0
    xp = np.asarray(x, dtype=np.float64) - x_0; yp = np.asarray(y, dtype=np.float64) - y_0
    rho = np.hypot(xp, yp)
    if lat_ts is not None and abs(abs(lat_ts) - 90.0) > 1e-9:
        pc = abs(np.radians(lat_ts)); s_ = np.sin(pc)
        tc = np.tan(0.5 * (np.pi / 2 - pc)) / (((1 - e * s_) / (1 + e * s_)) ** (0.5 * e) if e != 0 else 1.0)
        t = rho * tc / (a * np.cos(pc) / np.sqrt(1 - es * s_ ** 2))
    else:
        t = rho * np.sqrt((1 + e) ** (1 + e) * (1 - e) ** (1 - e)) / (2 * a * k_0)
    lat = sg * np.degrees(_phi_from_t(t, e))
    lon = lon_0 + np.degrees(np.arctan2(xp, -sg * yp))           # north: atan2(x, -y); south: atan2(x, y)
    lon = np.where(rho == 0, lon_0, lon)
    lon = np.where(lon > 180, lon - 360, np.where(lon < -180, lon + 360, lon))
    return lat, lon


def project_inverse(x, y, model_proj):
    kw = dict(model_proj)
    kind = kw.pop('proj', 'lcc')
    return stere_inverse(x, y, **kw) if kind == 'stere' else lcc_inverse(x, y, **kw)


def project_forward(lat, lon, model_proj):
    """(x, y) of geodetic points in the model CRS given as a dict: {'proj': 'stere', ...stere_forward keywords} or
    lcc_forward keywords (optionally with 'proj': 'lcc')."""
    kw = dict(model_proj)
    kind = kw.pop('proj', 'lcc')
    return stere_forward(lat, lon, **kw) if kind == 'stere' else lcc_forward(lat, lon, **kw)


def build_cube(xpts, ypts, zpts, interpolators, model_proj=None):
    """delay.py:196-216.  model_proj=None: model_crs == pts_crs (EPSG:4326 cube); else a dict of lcc_forward keyword
    arguments = the `transformPoints(yy, xx, ht, pts_crs, model_crs)` branch (delay.py:207-209) for an LCC model."""
    xx, yy = np.meshgrid(xpts, ypts)
    zpts = np.asarray(zpts)
    out = [np.zeros((zpts.size, len(ypts), len(xpts))) for _ in interpolators]
    if model_proj is not None:
        px, py = project_forward(yy, xx, model_proj)
        xx, yy = px, py
    for ii, ht in enumerate(zpts):
        pts = np.stack([yy, xx, np.full(yy.shape, ht)], axis=-1)
        for mm, intp in enumerate(interpolators):
            out[mm][ii, ...] = intp(pts)
    return out


def integrate_slice(model_zs, ray_lengths, low_xyzs, high_xyzs, nParts, interpolators, outSubs, model_proj=None):
    """delay.py:285-323: the sample loop for one slice (EPSG:4326 model cube, or an LCC one via model_proj)."""
    zmin = np.array(model_zs).min()
    zmax = np.array(model_zs).max()
    for zz, nparts in enumerate(nParts):
        fracs = np.linspace(0.0, 1.0, num=nparts)
        for findex, ff in enumerate(fracs):
            pts_xyz = low_xyzs[zz] + ff * (high_xyzs[zz] - low_xyzs[zz])
            lon, lat, h = ecef2lla(pts_xyz[..., 0], pts_xyz[..., 1], pts_xyz[..., 2])
            if model_proj is not None:      # ecef_to_model = ECEF -> geodetic -> model CRS (delay.py:253,295)
                lon, lat = project_forward(lat, lon, model_proj)
            pts = np.stack((lat, lon, h), axis=-1)
            if (pts[..., -1] < zmin).all():
                pts[..., -1] = zmin
            if (pts[..., -1] > zmax).all():
                pts[..., -1] = zmax
            wt = 0.5 if findex in [0, fracs.size - 1] else 1.0
            wt = wt * (ray_lengths[zz] * 1.0e-6 / (nparts - 1.0))
            for mm, out in enumerate(outSubs):
                val = interpolators[mm](pts)
                out += wt * val


def build_cube_ray(xpts, ypts, zpts, look_fn, interpolators, MAX_SEGMENT_LENGTH=1000.0,
                   MAX_TROPO_HEIGHT=_ZREF, nParts_override=None, return_nparts=False, model_proj=None):
    """delay.py:219-326 for an EPSG:4326 cube and EPSG:4326 query grid.

    `look_fn(ht, llh, xyz, yy) -> (ny,nx,3)` plays `los.getLookVectors` (delay.py:270).
    `nParts_override[hh]` (list per height level) replaces delay.py:283 - used to drive a shard with
    the whole-slice nParts."""
    model_zs = interpolators[0].grid[2]
    xx, yy = np.meshgrid(xpts, ypts)
    zpts = np.asarray(zpts)
    outputArrs = [np.zeros((zpts.size, len(ypts), len(xpts))) for _ in interpolators]
    all_nparts = []
    for hh, ht in enumerate(zpts):
        outSubs = [x[hh, ...] for x in outputArrs]
        llh = [xx, yy, np.full(yy.shape, ht)]
        xyz = np.stack(lla2ecef(llh[1], llh[0], llh[2]), axis=-1)
        LOS = look_fn(ht, llh, xyz, yy)
        ray_lengths, low_xyzs, high_xyzs = build_ray(model_zs, ht, xyz, LOS, MAX_TROPO_HEIGHT)
        if ray_lengths is None and ht == zpts[-1]:
            all_nparts.append(None)
            continue
        elif np.isnan(ray_lengths).all():
            raise ValueError('geo2rdr did not converge. Check orbit coverage')
        if nParts_override is not None:
            nParts = np.asarray(nParts_override[hh])
        else:
            nParts = nparts_from_lengths(ray_lengths, MAX_SEGMENT_LENGTH)
        all_nparts.append(nParts)
        integrate_slice(model_zs, ray_lengths, low_xyzs, high_xyzs, nParts, interpolators, outSubs, model_proj)
    if return_nparts:
        return outputArrs, all_nparts
    return outputArrs


def ray_levels_idx(model_zs, ht, zref):
    """ray_levels with the index zz of the model interval each entry comes from: list of (zz, low_ht, high_ht)."""
    model_zs = np.asarray(model_zs, dtype=np.float64)
    out = []
    lv = ray_levels(model_zs, ht, zref)
    # match every contributing (low, high) back to its interval: high_ht (before the zref clip) identifies it
    k = 0
    for zz in range(model_zs.size - 1):
        if k == len(lv):
            break
        high = model_zs[zz + 1] - (0.01 if model_zs[zz + 1] == model_zs[-1] else 0.0)
        low = model_zs[zz]
        if (high < ht) or (low >= zref):
            continue
        lo_c, hi_c = max(low, ht), min(high, zref)
        if np.abs(hi_c - lo_c) < 1.0:
            continue
        assert lv[k] == (float(lo_c), float(hi_c))
        out.append((zz, lv[k][0], lv[k][1]))
        k += 1
    assert k == len(lv)
    return out


def build_cube_ray_per_pixel(lat, lon, hts, LOS, interpolators, MAX_SEGMENT_LENGTH=1000.0, MAX_TROPO_HEIGHT=_ZREF, nParts_override=None):
    """Rays with their OWN origin heights (SURVEY 8d "c3b": no reference semantics).  The rule: the reference's
    slice algorithm (delay.py:256-323) ray by ray wherever it is per ray - build_ray with the ray's height, its first contributing
    interval fixing cos_factor - and batch-level wherever the reference reduces over the slice: nParts[zz] = ceil(max over the
    rays interval zz contributes to / MAX_SEGMENT_LENGTH) + 1 (delay.py:283), and the all-pixels z-clamp (delay.py:306-311) asked
    about every ray's own first / last sample.  Built from the pinned slice functions (build_ray, scipy-RGI restatement), one ray
    at a time - small inputs only; oracle_c.build_cube_ray_per_pixel is the fast form.
    lat, lon, hts: 1-D arrays; LOS (n, 3).  Returns (wet[n], hydro[n], nparts[nz-1] indexed by model interval)."""
    model_zs = np.asarray(interpolators[0].grid[2], dtype=np.float64)
    lat = np.asarray(lat, dtype=np.float64).ravel(); lon = np.asarray(lon, dtype=np.float64).ravel()
    hts = np.asarray(hts, dtype=np.float64)
    hts = hts.ravel() if hts.size == lat.size else np.broadcast_to(hts, lat.shape).ravel()
    LOS = np.asarray(LOS, dtype=np.float64).reshape(-1, 3)
    n = lat.size
    M = model_zs.size - 1
    rays = []
    maxlen = np.zeros(M)
    for i in range(n):
        xyz = np.stack(lla2ecef(lat[i:i + 1], lon[i:i + 1], hts[i:i + 1]), axis=-1)
        lens, lows, highs = build_ray(model_zs, hts[i], xyz, LOS[i:i + 1], MAX_TROPO_HEIGHT)
        idx = [zz for zz, _, _ in ray_levels_idx(model_zs, hts[i], MAX_TROPO_HEIGHT)]
        rays.append((idx, lens, lows, highs))
        for k, zz in enumerate(idx):
            L = lens[k][0]
            if not np.isnan(maxlen[zz]) and (np.isnan(L) or L > maxlen[zz]):
                maxlen[zz] = L
    if nParts_override is not None:
        nparts = np.asarray(nParts_override)
    else:
        if np.isnan(maxlen).any():
            raise ValueError('some ray lengths are NaN: the number of integration parts (delay.py:283) is undefined')
        nparts = np.where(maxlen > 0, np.ceil(maxlen / MAX_SEGMENT_LENGTH) + 1, 0).astype(int)
    zmin, zmax = model_zs.min(), model_zs.max()
    firsts = [ecef2lla(*(r[2][0][0]))[2] for r in rays if r[0]]
    lasts = [ecef2lla(*(r[3][-1][0]))[2] for r in rays if r[0]]
    clamp_lo = bool(firsts) and all(h < zmin for h in firsts)
    clamp_hi = bool(lasts) and all(h > zmax for h in lasts)
    wet = np.zeros(n); hyd = np.zeros(n)
    for i, (idx, lens, lows, highs) in enumerate(rays):
        for k, zz in enumerate(idx):
            fracs = np.linspace(0.0, 1.0, num=int(nparts[zz]))
            for findex, ff in enumerate(fracs):
                pts_xyz = lows[k] + ff * (highs[k] - lows[k])
                plon, plat, ph = ecef2lla(pts_xyz[..., 0], pts_xyz[..., 1], pts_xyz[..., 2])
                pts = np.stack((plat, plon, ph), axis=-1)
                if clamp_lo and k == 0 and findex == 0:
                    pts[..., -1] = zmin
                if clamp_hi and k == len(idx) - 1 and findex == fracs.size - 1:
                    pts[..., -1] = zmax
                wt = 0.5 if findex in [0, fracs.size - 1] else 1.0
                wt = wt * (lens[k] * 1.0e-6 / (nparts[zz] - 1.0))
                wet[i] += (wt * interpolators[0](pts))[0]
                hyd[i] += (wt * interpolators[1](pts))[0]
    return wet, hyd, nparts


def points_from_cube(lats, lons, hgts, xpts, ypts, zpts, wet_cube, hydro_cube):
    """delay.py:110-121: second-stage interpolation of an output delay cube (z,y,x) to stations;
    `getInterpolators(ds,'ztd')` on the output Dataset (delayFcns.py:37-41: kind!='total' picks
    'wet'/'hydro', transposed to (y,x,z))."""
    ifW, ifH = getInterpolators(xpts, ypts, zpts, wet_cube, hydro_cube)
    pnts = np.stack([lats, lons, hgts], axis=-1)
    return ifW(pnts), ifH(pnts)


# ----------------------------------------------------------------------------------------------
# temporal blend (cli/raider.py:817-819 + :877-888)
# ----------------------------------------------------------------------------------------------
def blend_cubes(w1, a1, w2, a2):
    """`ds_out[var] = sum([wgt * ds[var] ...])` (cli/raider.py:817-819) = 0 + w1*a1 + w2*a2.

    Under the reference's pinned numpy<2 (environment.yml:30) a float64 *scalar* weight times a
    float32 array is computed in float32 (value-based casting), so `wet`/`hydro` stay f32 and
    `*_total` stay f64.  Restated explicitly so the result does not depend on the numpy in use."""
    dt_ = a1.dtype
    return (dt_.type(w1) * a1 + dt_.type(w2) * a2).astype(dt_)


def time_weights(t, t1, t2):
    """cli/raider.py:877-888 (two-epoch linear weights), times in seconds."""
    span = abs(t2 - t1)
    return 1 - abs(t - t1) / span, 1 - abs(t2 - t) / span


# ----------------------------------------------------------------------------------------------
# azimuth-time-grid temporal weighting (s1_azimuth_timing.py)
# ----------------------------------------------------------------------------------------------
def n_closest_datetimes(ref_time, n_target_times, time_step_hours):
    """s1_azimuth_timing.py:204-266 without pandas: model times (multiples of the step since midnight) around ref_time,
    ordered by distance (ties: earlier first), first n."""
    import datetime as _dt
    if 24 % time_step_hours != 0:
        raise ValueError('The time step does not evenly divide 24 hours')
    step = _dt.timedelta(hours=time_step_hours)
    day0 = _dt.datetime(ref_time.year, ref_time.month, ref_time.day)

    def floor(t):
        return day0 + ((t - day0) // step) * step

    def ceil(t):
        f = floor(t)
        return f if f == t else f + step
    found = []
    for k in range(int(np.ceil(n_target_times / 2))):
        for t in {floor(ref_time - k * step), ceil(ref_time + k * step)}:
            found.append(t)
    found = sorted(found, key=lambda t: (abs(ref_time - t), t))
    return found[:n_target_times]


def times_for_azimuth_interpolation(ref_time, time_step_hours, buffer_in_seconds=300):
    """s1_azimuth_timing.py:269-323: the 3 closest model times that lie within one step (+ buffer) of ref_time."""
    bound = time_step_hours * 3600 + buffer_in_seconds
    return [t for t in n_closest_datetimes(ref_time, 3, time_step_hours) if abs((ref_time - t).total_seconds()) < bound]


def inverse_time_weights(az_s, dates_s, window_s=None, regularizer=1e-9):
    """s1_azimuth_timing.py:326-399 with times in seconds on a common epoch: w_i = m_i / (|t - d_i| + reg) normalised over
    the dates, m_i = |t - d_i| <= window.  Returns (len(dates),) + az_s.shape; voxels with no date in the window are NaN."""
    dates_s = [float(d) for d in dates_s]
    if len(set(dates_s)) != len(dates_s):
        raise ValueError('Dates provided must be unique')
    if not dates_s:
        raise ValueError('No dates provided')
    if window_s is None:
        window_s = min(abs(d - dates_s[0]) for d in dates_s[1:])
    az_s = np.asarray(az_s, dtype=np.float64)
    diffs = [np.abs(az_s - d) for d in dates_s]
    masked = [(1.0 / (df + regularizer)) * (df <= window_s).astype(int) for df in diffs]
    if all((df <= window_s).sum() == 0 for df in diffs):
        raise ValueError('No dates provided are within temporal window')
    total = np.sum(np.stack(masked, axis=-1), axis=-1)
    with np.errstate(invalid='ignore', divide='ignore'):
        return np.stack([m / total for m in masked])


def azimuth_time_grid(st, sp, sv, lat, lon, hgt):
    """s1_azimuth_timing.py:90-147 with this build's zero-Doppler solver (isce3 is not available: unpinned):
    seconds (orbit time scale) of the zero-Doppler epoch PLUS the one-way range delay sr/c (:138-139), truncated to
    milliseconds like the datetime64[ms] array the reference fills."""
    xyz = np.stack(lla2ecef(lat, lon, hgt), axis=-1)
    _, t, rg = orbit_look_vectors(st, sp, sv, xyz, threshold=1.0e-7, maxiter=100)
    return np.floor((t + rg / 299792458.0) * 1e3) / 1e3


def combine_weighted(weights, fields):
    """cli/raider.py:817-819 with per-voxel weights: sum([wgt * ds[var] ...]) = ((0 + w0 f0) + w1 f1) + ...  in f64."""
    out = 0
    for w, f in zip(weights, fields):
        out = out + w * f
    return out


def gunw_phase(delay, wavelength):
    """aria/calcGUNW.py:54-59: `ds['wet'] * phase2range`, phase2range = (-4 * np.pi) / float(wavelength).  A Python-float
    factor keeps the array's dtype (float32 delays stay float32)."""
    phase2range = (-4 * np.pi) / float(wavelength)
    return np.asarray(delay) * phase2range


# ----------------------------------------------------------------------------------------------
# native extension restatements
# ----------------------------------------------------------------------------------------------
def _bisect(grid, x):
    """interpolate.h:23-38: first index with x < grid[i]  (== searchsorted side='right')."""
    return np.searchsorted(grid, x, side='right')


def native_interpolate(points, values, interp_points, fill_value=None):
    """`RAiDER.interpolate.interpolate` (module.cpp:26-294; interpolate.h:78-118,
    interpolate.cpp:18-258): N-D linear, hi = upper_bound index; with fill: hi<1 or hi>N-1 -> fill
    (so a query ON the last node is filled); without fill: hi clamped to [1,N-1] (extrapolate).
    value = sum(corner * prod(dist)) / prod(dx)   (1-D: y0 + slope*(x-x0))."""
    points = [np.asarray(p, dtype=np.float64) for p in points]
    values = np.asarray(values, dtype=np.float64)
    q = np.asarray(interp_points, dtype=np.float64)
    nd = len(points)
    n = q.shape[0]
    filled = np.zeros(n, dtype=bool)
    lo, hi = [], []
    for d, g in enumerate(points):
        h = _bisect(g, q[:, d])
        if fill_value is not None:
            filled |= (h < 1) | (h > g.size - 1)
        h = np.clip(h, 1, g.size - 1)
        hi.append(h)
        lo.append(h - 1)
    if nd == 1:
        g = points[0]
        x0, x1 = g[lo[0]], g[hi[0]]
        y0, y1 = values[lo[0]], values[hi[0]]
        slope = (y1 - y0) / (x1 - x0)
        out = y0 + slope * (q[:, 0] - x0)
    else:
        d0 = [q[:, d] - points[d][lo[d]] for d in range(nd)]   # dist to lower
        d1 = [points[d][hi[d]] - q[:, d] for d in range(nd)]   # dist to upper
        vol = np.ones(n)
        for d in range(nd):
            vol = vol * (points[d][hi[d]] - points[d][lo[d]])
        if nd == 2:
            z = lambda a, b: values[a, b]
            out = (d1[0] * (z(lo[0], lo[1]) * d1[1] + z(lo[0], hi[1]) * d0[1]) +
                   d0[0] * (z(hi[0], lo[1]) * d1[1] + z(hi[0], hi[1]) * d0[1])) / vol
        elif nd == 3:
            w = lambda a, b, c: values[a, b, c]
            out = (d1[0] * (d1[1] * (d1[2] * w(lo[0], lo[1], lo[2]) + d0[2] * w(lo[0], lo[1], hi[2])) +
                            d0[1] * (d1[2] * w(lo[0], hi[1], lo[2]) + d0[2] * w(lo[0], hi[1], hi[2]))) +
                   d0[0] * (d1[1] * (d1[2] * w(hi[0], lo[1], lo[2]) + d0[2] * w(hi[0], lo[1], hi[2])) +
                            d0[1] * (d1[2] * w(hi[0], hi[1], lo[2]) + d0[2] * w(hi[0], hi[1], hi[2])))) / vol
        else:
            out = np.zeros(n)
            for j in range(1 << nd):
                # interpolate.cpp:238-250: bit `dim` of j selects hi/lo of dimension dim
                index = tuple(hi[d] if (j >> d) & 1 else lo[d] for d in range(nd))
                term = values[index]
                for d in range(nd):
                    term = term * (d0[d] if (j >> d) & 1 else d1[d])
                out = out + term
            out = out / vol
    if fill_value is not None:
        out = np.where(filled, fill_value, out)
    return out


def native_interpolate_along_axis(points, values, interp_points, axis=-1, fill_value=None):
    """`RAiDER.interpolate.interpolate_along_axis` (module.cpp:296-493, interpolate.cpp:260-332):
    independent interpolate_1d (interpolate.h:78-118) along `axis` of same-shaped arrays."""
    points = np.asarray(points, dtype=np.float64)
    values = np.asarray(values, dtype=np.float64)
    q = np.asarray(interp_points, dtype=np.float64)
    axis = axis % points.ndim
    P = np.moveaxis(points, axis, -1)
    V = np.moveaxis(values, axis, -1)
    Q = np.moveaxis(q, axis, -1)
    lead = P.shape[:-1]
    P2 = P.reshape(-1, P.shape[-1])
    V2 = V.reshape(-1, V.shape[-1])
    Q2 = Q.reshape(-1, Q.shape[-1])
    out = np.empty_like(Q2)
    for r in range(P2.shape[0]):
        out[r] = native_interpolate((P2[r],), V2[r], Q2[r][:, None], fill_value)
    return np.moveaxis(out.reshape(lead + (Q.shape[-1],)), -1, axis)


def makePoints(max_len, Rays_SP, Rays_SLV, stepSize):
    """makePoints.pyx:15-148 (0-D..3-D share one formula):
    ray[..., k3, k4] = SP[..., k3] + basespace[k4]*SLV[..., k3], basespace = arange(0, max_len+step, step),
    Npts = int(max_len//step) + (1 if max_len % step != 0 else 0)."""
    SP = np.asarray(Rays_SP, dtype=np.float64)
    SLV = np.asarray(Rays_SLV, dtype=np.float64)
    if max_len % stepSize != 0:
        Npts = int(max_len // stepSize) + 1
    else:
        Npts = int(max_len // stepSize)
    basespace = np.arange(0, max_len + stepSize, stepSize)[:Npts]
    return SP[..., :, None] + basespace * SLV[..., :, None]


# ----------------------------------------------------------------------------------------------
# cube producer  (models/weatherModel.py:235-262: what turns model-level (p, t, q|rh, z) columns into the four fields the
# delay path reads).  Pinned by golden g10 (the reference's own WeatherModel run on synthetic columns).
# ----------------------------------------------------------------------------------------------
def find_svp(t):
    """models/weatherModel.py:750-780 (saturation vapour pressure, Pa, returned as float32)."""
    t = np.asarray(t)
    t1, t2 = 273.15, 250.15
    tref = t - t1
    wgt = (t - t2) / (t1 - t2)
    svpw = 6.1121 * np.exp((17.502 * tref) / (240.97 + tref))
    svpi = 6.1121 * np.exp((22.587 * tref) / (273.86 + tref))
    svp = svpi + (svpw - svpi) * wgt ** 2
    svp = np.where(t > t1, svpw, svp)
    svp = np.where(t < t2, svpi, svp)
    return (svp * 100).astype(np.float32)


def fillna_columns(a, fill_value=0.0):
    """interpolator.py:110-130 `fillna3D` (pandas interpolate(axis=1, limit_direction='backward') along the last axis):
    leading NaNs <- first valid value, interior NaN runs <- linear in the INDEX, trailing NaNs <- fill_value."""
    a = np.asarray(a)
    out = a.copy()
    flat = out.reshape(-1, a.shape[-1])
    idx = np.arange(a.shape[-1], dtype=np.float64)
    for row in flat:
        ok = ~np.isnan(row)
        if not ok.any():
            row[:] = fill_value
            continue
        last = np.nonzero(ok)[0][-1]
        filled = np.interp(idx, idx[ok], row[ok].astype(np.float64))
        row[:last + 1] = filled[:last + 1].astype(a.dtype)
        row[last + 1:] = fill_value
    return out


def cube_from_model_levels(zs, p, t, hum, humidity_type, new_z, k1=0.776, k2=0.233, k3=3.75e3, zmin=_ZMIN,
                           R_v=461.524, R_d=287.06):
    """models/weatherModel.py:235-262 on arrays shaped (A, B, nlev) with per-column ascending heights `zs`:
    _find_e (:332-353) -> _uniform_in_z (:603-629, native interpolate_along_axis with NaN fill, cast to f32) ->
    _checkForNans (:631-635) -> refractivities (:355-361) -> _adjust_grid (:371-387, pad a level at zmin) -> _getZTD
    (:389-403).  Returns dict(zs, t, p, e, wet, hydro (f32, (A,B,nz)), wet_total, hydro_total (f64))."""
    svp = find_svp(t)
    if humidity_type == 'q':
        w = hum / (1 - hum)
        e = w * R_v * (p - svp) / R_d
    elif humidity_type == 'rh':
        e = hum / 100 * svp
    else:
        raise RuntimeError('Not a valid humidity type')
    new_z = np.asarray(new_z, dtype=np.float64)
    new3 = np.broadcast_to(new_z, zs.shape[:2] + (new_z.size,))
    tu = native_interpolate_along_axis(zs, t, new3, axis=2, fill_value=np.nan).astype(np.float32)
    pu = native_interpolate_along_axis(zs, p, new3, axis=2, fill_value=np.nan).astype(np.float32)
    eu = native_interpolate_along_axis(zs, e, new3, axis=2, fill_value=np.nan).astype(np.float32)
    pf = fillna_columns(pu)
    tf = fillna_columns(tu, fill_value=1e16)
    ef = fillna_columns(eu)
    wet = (np.float32(k2) * ef / tf + np.float32(k3) * ef / tf ** 2).astype(np.float32)
    hydro = (np.float32(k1) * pf / tf).astype(np.float32)
    out_z = new_z
    if zmin < np.nanmin(new_z):                       # _adjust_grid: new lowest level at zmin holding each column's lowest value
        out_z = np.insert(new_z, 0, zmin)
        pad = lambda v: np.concatenate((v[:, :, :1], v), axis=2)
        pf, tf, ef, wet, hydro = pad(pf), pad(tf), pad(ef), pad(wet), pad(hydro)

    def ztd(f):
        tot = np.zeros(f.shape)
        for level in range(f.shape[2]):
            y = f[..., level:]
            d = np.diff(out_z[level:])
            tot[..., level] = 1e-6 * (d * (y[..., 1:] + y[..., :-1]) / 2.0).sum(-1)
        return tot
    return dict(zs=out_z, t=tf, p=pf, e=ef, wet=wet, hydro=hydro, wet_total=ztd(wet), hydro_total=ztd(hydro),
                t_u=tu, p_u=pu, e_u=eu, e_levels=e)


# ----------------------------------------------------------------------------------------------
# synthetic workloads - shared by tests and bench so GPU and CPU see the
# same seeded inputs
# ----------------------------------------------------------------------------------------------
def read_ecmwf_model_level_file(path, ll_bounds=None):
    """models/ecmwf.py:305-337 `_makeDataCubes` + :58-79 for a raw ERA-5 / HRES model-level file (NetCDF-3 as the CDS writes
    it: packed int16 `z, t, q, lnsp` on (time, level, latitude, longitude)): CF-decode to float32 (what xarray did for int16
    data when the reference's test cubes were made: `data.astype(float32) * scale_factor + add_offset`, in float32), wrap the
    longitudes to [-180, 180), keep the nodes inside ll_bounds = (S, N, W, E), make latitude / longitude ascending.
    Returns dict(lats, lons, z (ny,nx), lnsp (ny,nx), t, q (nlev,ny,nx))."""
    from scipy.io import netcdf_file
    with netcdf_file(str(path), 'r', mmap=False) as f:
        def dec(name):
            v = f.variables[name]
            raw = np.array(v.data)
            out = raw.astype(np.float32)
            out *= np.float32(v.scale_factor)
            out += np.float32(v.add_offset)
            out[raw == v._FillValue] = np.nan
            return out
        z, t, q, lnsp = (np.squeeze(dec(k)) for k in ('z', 't', 'q', 'lnsp'))
        lats = np.array(f.variables['latitude'].data, dtype=np.float32)
        lons = np.array(f.variables['longitude'].data, dtype=np.float32)
    lons = ((lons + 180) % 360) - 180                                              # ecmwf.py:314
    z, lnsp = z[0], lnsp[0]                                                        # :322,325 (the surface fields sit on level 1)
    if ll_bounds is not None:
        S, N, W, E = ll_bounds
        my = (S <= lats) & (N >= lats); mx = (W <= lons) & (E >= lons)             # :317-319
        lats, lons = lats[my], lons[mx]
        z, lnsp, t, q = z[my][:, mx], lnsp[my][:, mx], t[:, my][:, :, mx], q[:, my][:, :, mx]
    if lats[0] > lats[1]:                                                          # :63-68
        z, lnsp, t, q, lats = z[::-1], lnsp[::-1], t[:, ::-1], q[:, ::-1], lats[::-1]
    if lons[0] > lons[1]:                                                          # :70-75
        z, lnsp, t, q, lons = z[..., ::-1], lnsp[..., ::-1], t[..., ::-1], q[..., ::-1], lons[::-1]
    return dict(lats=lats, lons=lons, z=np.ascontiguousarray(z), lnsp=np.ascontiguousarray(lnsp), t=np.ascontiguousarray(t), q=np.ascontiguousarray(q))


def ecmwf_model_levels(z_surf, lnsp, t, q, lats, a, b, R_d=287.06, g0=9.80665, dtype=np.float32):
    """utilFcns.calcgeoh (:781-859) + geo_to_ht (:378-410, with _get_g_ll :351-353 and get_Re :356-376) + the re-ordering of
    ecmwf.py:92-110: hybrid-level pressures, geopotential integrated upwards from the surface, geopotential height, geometric
    height.  dtype=float32 is what the reference really ran when its test cubes were written: float32 arrays decoded from the
    packed file, Python-float constants, NumPy-1 value-based casting, so every operation is float32.  That evaluation is
    ill-conditioned - dlogP = log(P1) - log(P0) and alpha = 1 - P0/(P1-P0) dlogP lose 3-4 digits - and lands up to 2.4 m from
    the float64 evaluation of the same formulas (dtype=float64), with a last-bit change in log() moving a height by metres.
    t, q: (nlev, ny, nx) with level 1 = model top; returns (p, hgt) as (ny, nx, nlev) `dtype`, bottom level first."""
    ft = np.dtype(dtype).type
    t = np.asarray(t, dtype=ft); q = np.asarray(q, dtype=ft)
    z_surf = np.asarray(z_surf, dtype=ft); lnsp = np.asarray(lnsp, dtype=ft)
    nlev = t.shape[0]
    a = [float(v) for v in a]; b = [float(v) for v in b]
    if len(a) != nlev + 1 or len(b) != nlev + 1:
        raise ValueError(f'I have here a model with {nlev} levels, but parameters a and b have lengths {len(a)} and {len(b)} '
                         'respectively. Of course, these three numbers should be equal.')
    pres = np.zeros_like(t); gh = np.zeros_like(t)
    sp = np.exp(lnsp)
    z_h = 0
    for lev in range(nlev, 0, -1):
        il = lev - 1
        tl = t[il] * (1 + ft(0.609133) * q[il])                                    # moist temperature
        ph = ft(a[lev - 1]) + (ft(b[lev - 1]) * sp); ph1 = ft(a[lev]) + (ft(b[lev]) * sp)
        pres[il] = ph
        if lev == 1:
            dlogp = np.log(ph1 / ft(0.1)); alpha = ft(np.log(2))
        else:
            dlogp = np.log(ph1) - np.log(ph)
            alpha = 1 - ((ph / (ph1 - ph)) * dlogp)
        trd = tl * ft(R_d)
        z_f = z_h + trd * alpha + z_surf
        gh[il] = z_f / ft(g0)
        z_h = z_h + trd * dlogp
    hgt = gh.transpose(1, 2, 0)
    latf = np.broadcast_to(np.asarray(lats, dtype=ft)[:, None, None], hgt.shape)
    c2 = np.cos(np.radians(2 * latf))
    g_ll = ft(9.80616) * (1 - ft(0.002637) * c2 + ft(0.0000059) * c2 ** 2)
    cl, sl = np.cos(np.radians(latf)), np.sin(np.radians(latf))
    re = np.sqrt(1 / (((cl ** 2) / ft(6378137 ** 2)) + ((sl ** 2) / ft(6356752 ** 2))))
    h = (hgt * re) / (g_ll / ft(g0) * re - hgt)
    assert h.dtype == ft and pres.dtype == ft
    return np.flip(pres.transpose(1, 2, 0), axis=2).copy(), np.flip(h, axis=2).copy()


def ztd_totals(field_yxz, zs):
    """weatherModel.py:389-403 `_getZTD` on a (y, x, z) field: total[..., l] = 1e-6 * trapz(field[..., l:], zs[l:])."""
    f = np.asarray(field_yxz)
    out = np.zeros(f.shape, dtype=np.float64)
    for lev in range(zs.size):
        y = f[..., lev:]
        d = np.diff(zs[lev:])
        out[..., lev] = 1e-6 * np.sum(d * (y[..., 1:] + y[..., :-1]) / 2.0, axis=-1)
    return out


def synthetic_cube(ny, nx, nz, seed=0, ztop=41000.0, y0=30.0, y1=36.0, x0=-121.0, x1=-113.0):
    """SURVEY §8(d) "Synthetic cube": returns dict(xs, ys, zs, wet, hydro (z,y,x) f32,
    wet_total, hydro_total (z,y,x) f64)."""
    ys = np.linspace(y0, y1, ny)
    xs = np.linspace(x0, x1, nx)
    zs = np.round(-100 + ztop * np.linspace(0, 1, nz) ** 2, 3)
    rng = np.random.default_rng(seed)
    g_h = rng.standard_normal((ny, nx))
    g_w = rng.standard_normal((ny, nx))
    z3 = zs[:, None, None]
    hydro = (270.0 * np.exp(-z3 / 8000.0) * (1 + 0.01 * g_h[None])).astype(np.float32)
    wet = (60.0 * np.exp(-z3 / 2000.0) * (1 + 0.1 * g_w[None])).astype(np.float32)

    def totals(f):
        # weatherModel.py:389-403 `_getZTD`: total[l] = 1e-6 * trapz(f[l:], zs[l:])
        f = f.astype(np.float64)
        seg = 0.5 * (f[1:] + f[:-1]) * np.diff(zs)[:, None, None]
        cum = np.concatenate([np.cumsum(seg[::-1], axis=0)[::-1], np.zeros((1,) + f.shape[1:])], axis=0)
        return 1e-6 * cum
    return dict(xs=xs, ys=ys, zs=zs, wet=wet, hydro=hydro, wet_total=totals(wet), hydro_total=totals(hydro))
